/* tests/nav_probe.c - test helper: sends one mouse-move navigation event upstream into a running
 *   videotestsrc(A) -> vfhipcompositor.sink_0 ; videotestsrc(B) -> vfhipcompositor.sink_1 (xpos, ypos, width, height) ; -> fakesink
 * pipeline and prints, for every source that received it, the pointer coordinates it saw.
 * usage: nav_probe <pointer_x> <pointer_y> */
#include <gst/gst.h>
#include <stdio.h>
#include <stdlib.h>

static GstPadProbeReturn
on_event (GstPad * pad, GstPadProbeInfo * info, gpointer name)
{
  GstEvent *ev = GST_PAD_PROBE_INFO_EVENT (info);
  if (GST_EVENT_TYPE (ev) == GST_EVENT_NAVIGATION) {
    gdouble x = -1, y = -1;
    const GstStructure *s = gst_event_get_structure (ev);
    gst_structure_get_double (s, "pointer_x", &x);
    gst_structure_get_double (s, "pointer_y", &y);
    printf ("%s %.3f %.3f\n", (const char *) name, x, y);
    fflush (stdout);
  }
  (void) pad;
  return GST_PAD_PROBE_OK;
}

int
main (int argc, char **argv)
{
  GstElement *p, *a, *b, *sink;
  GstPad *pad;
  GstStateChangeReturn rc;
  GError *err = NULL;
  gst_init (&argc, &argv);
  if (argc < 3) return 2;
  p = gst_parse_launch ("vfhipcompositor name=c sink_1::xpos=200 sink_1::ypos=100 sink_1::width=160 sink_1::height=60 ! fakesink name=out sync=false "
      "videotestsrc name=a is-live=true ! video/x-raw,format=BGRA,width=320,height=240,framerate=30/1 ! c.sink_0 "
      "videotestsrc name=b is-live=true ! video/x-raw,format=NV12,width=80,height=120,framerate=30/1 ! c.sink_1", &err);
  if (!p) { fprintf (stderr, "parse: %s\n", err->message); return 1; }
  a = gst_bin_get_by_name (GST_BIN (p), "a"); b = gst_bin_get_by_name (GST_BIN (p), "b"); sink = gst_bin_get_by_name (GST_BIN (p), "out");
  pad = gst_element_get_static_pad (a, "src"); gst_pad_add_probe (pad, GST_PAD_PROBE_TYPE_EVENT_UPSTREAM, on_event, "a", NULL); gst_object_unref (pad);
  pad = gst_element_get_static_pad (b, "src"); gst_pad_add_probe (pad, GST_PAD_PROBE_TYPE_EVENT_UPSTREAM, on_event, "b", NULL); gst_object_unref (pad);
  gst_element_set_state (p, GST_STATE_PLAYING);
  rc = gst_element_get_state (p, NULL, NULL, 20 * GST_SECOND);
  if (rc == GST_STATE_CHANGE_FAILURE) { fprintf (stderr, "state change failed\n"); return 1; }
  g_usleep (300000);                                    /* a few frames: caps negotiated, output size known */
  pad = gst_element_get_static_pad (sink, "sink");
  gst_pad_push_event (pad, gst_event_new_navigation (gst_structure_new ("application/x-gst-navigation", "event", G_TYPE_STRING, "mouse-move",
              "pointer_x", G_TYPE_DOUBLE, atof (argv[1]), "pointer_y", G_TYPE_DOUBLE, atof (argv[2]), NULL)));
  gst_object_unref (pad);
  g_usleep (100000);
  gst_element_set_state (p, GST_STATE_NULL);
  gst_object_unref (a); gst_object_unref (b); gst_object_unref (sink); gst_object_unref (p);
  return 0;
}
