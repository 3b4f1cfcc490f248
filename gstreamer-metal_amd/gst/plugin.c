/* gst/plugin.c — registers the vfhip elements (reference: src/plugin.m:36-62, plugin "vfmetal").
 * Element names follow BASELINE.json (vfhipconvertscale, ...).  Built with -DVFHIP_REGISTER_VFMETAL_NAMES the same
 * types are also registered under the reference's vfmetal* names so existing pipeline strings run unchanged. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include "gstvfhip.h"

GST_DEBUG_CATEGORY (gst_vfhip_debug);

#ifndef PACKAGE
#define PACKAGE "gst-vfhip"
#endif

static gboolean
plugin_init (GstPlugin * plugin)
{
  gboolean ok = TRUE;
  GST_DEBUG_CATEGORY_INIT (gst_vfhip_debug, "vfhip", 0, "MI355X (HIP) video processing elements");
  ok &= gst_vfhip_convertscale_register (plugin);
  ok &= gst_vfhip_videofilter_register (plugin);
  ok &= gst_vfhip_deinterlace_register (plugin);
  ok &= gst_vfhip_compositor_register (plugin);
  ok &= gst_vfhip_transform_register (plugin);
  ok &= gst_vfhip_overlay_register (plugin);
  return ok;
}

GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, vfhip,
    "HIP-accelerated (AMD MI355X) video processing elements", plugin_init, "1.0.0", "LGPL", "GstVfHip",
    "https://github.com/visioforge/gstreamer-metal")
