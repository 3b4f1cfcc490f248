/* gst/gstvfhipmemory.c — device-resident GstMemory and the `memory:HIPMemory` caps feature.
 *
 * The reference bounces every frame through host memory between elements (two CPU copies per element per frame,
 * common/vfmetaltextureutil.m:108 and common/vfmetalyuvoutput.m:138-176; chains in tests/test-multi-element.sh).  On a
 * discrete GPU that is a PCIe round trip per element.  vfhip elements therefore also negotiate
 * `video/x-raw(memory:HIPMemory)`: buffers of that kind hold ONE GstMemory whose payload lives in HBM, laid out like
 * the system-memory frame of the same GstVideoInfo.  A vfhip element maps it with GST_MAP_VFHIP and gets the device
 * pointer (VFHIP_FRAME_FLAG_DEVICE in the C ABI: no upload / download on that side); anything else that maps it for
 * CPU access (filesink, videoconvert ...) transparently gets a host shadow copy, downloaded on map(READ) and uploaded
 * on unmap(WRITE), so the memory is safe in any pipeline.  Upstream elements are synchronous (the frame is complete
 * when the buffer is pushed), so no cross-stream fencing is needed here. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/base/gstbasetransform.h>
#include <gst/video/gstvideofilter.h>
#include "gstvfhip.h"

#define GST_CAT_DEFAULT gst_vfhip_debug
#define VFHIP_DEVICE_MEMORY_TYPE "VfHipDeviceMemory"

typedef struct
{
  GstMemory mem;
  gpointer dev;                 /* device allocation (HBM) */
  gpointer shadow;              /* host copy for CPU mappings, allocated on demand */
  gint device;
  GMutex lock;
} VfHipDeviceMemory;

typedef struct
{
  GstAllocator parent;
  gint device;
} GstVfHipDeviceAllocator;
typedef struct
{
  GstAllocatorClass parent_class;
} GstVfHipDeviceAllocatorClass;

G_DEFINE_TYPE (GstVfHipDeviceAllocator, gst_vfhip_device_allocator, GST_TYPE_ALLOCATOR);

static GstMemory *
device_alloc (GstAllocator * allocator, gsize size, GstAllocationParams * params)
{
  GstVfHipDeviceAllocator *self = (GstVfHipDeviceAllocator *) allocator;
  VfHipDeviceMemory *m;
  /* +256: the kernels' vector loads may touch up to 16 bytes past the last row (like libvfhip's own staging images) */
  const gsize maxsize = size + params->prefix + params->padding + 256;
  gpointer dev = vfhip_device_malloc (self->device, maxsize);
  if (!dev) {
    GST_ERROR ("device allocation of %" G_GSIZE_FORMAT " bytes failed: %s", maxsize, vfhip_last_error_string ());
    return NULL;
  }
  m = g_slice_new0 (VfHipDeviceMemory);
  m->dev = dev;
  m->device = self->device;
  g_mutex_init (&m->lock);
  /* hipMalloc returns 256-byte aligned blocks; the payload starts at offset 0 (prefix / padding only add slack at the end) */
  gst_memory_init (GST_MEMORY_CAST (m), params->flags | GST_MEMORY_FLAG_NO_SHARE, allocator, NULL, maxsize, 255, 0, size);
  return GST_MEMORY_CAST (m);
}

static void
device_free (GstAllocator * allocator, GstMemory * mem)
{
  VfHipDeviceMemory *m = (VfHipDeviceMemory *) mem;
  (void) allocator;
  vfhip_device_free (m->device, m->dev);
  g_free (m->shadow);
  g_mutex_clear (&m->lock);
  g_slice_free (VfHipDeviceMemory, m);
}

static gpointer
device_map_full (GstMemory * mem, GstMapInfo * info, gsize maxsize)
{
  VfHipDeviceMemory *m = (VfHipDeviceMemory *) mem;
  gpointer ret;
  if (info->flags & GST_MAP_VFHIP)
    return m->dev;
  g_mutex_lock (&m->lock);
  if (!m->shadow)
    m->shadow = g_malloc0 (maxsize);
  /* ALWAYS bring the device contents into the shadow, also for a write-only mapping: unmap (WRITE) uploads the whole shadow,
   * so any byte the mapper does not write (a partial write, the padding of a stride) must hold what the device holds — an
   * uninitialised or stale shadow would overwrite valid frame data in HBM */
  if (vfhip_memcpy_d2h (m->device, m->shadow, m->dev, maxsize) != VFHIP_OK) {
    GST_ERROR ("download for a CPU mapping failed: %s", vfhip_last_error_string ());
    g_mutex_unlock (&m->lock);
    return NULL;
  }
  ret = m->shadow;
  g_mutex_unlock (&m->lock);
  return ret;
}

static void
device_unmap_full (GstMemory * mem, GstMapInfo * info)
{
  VfHipDeviceMemory *m = (VfHipDeviceMemory *) mem;
  if (info->flags & GST_MAP_VFHIP)
    return;
  if (info->flags & GST_MAP_WRITE) {
    g_mutex_lock (&m->lock);
    if (m->shadow && vfhip_memcpy_h2d (m->device, m->dev, m->shadow, mem->maxsize) != VFHIP_OK)
      GST_ERROR ("upload after a CPU mapping failed: %s", vfhip_last_error_string ());
    g_mutex_unlock (&m->lock);
  }
}

static void
gst_vfhip_device_allocator_class_init (GstVfHipDeviceAllocatorClass * klass)
{
  GST_ALLOCATOR_CLASS (klass)->alloc = device_alloc;
  GST_ALLOCATOR_CLASS (klass)->free = device_free;
}

static void
gst_vfhip_device_allocator_init (GstVfHipDeviceAllocator * self)
{
  GstAllocator *a = GST_ALLOCATOR_CAST (self);
  a->mem_type = VFHIP_DEVICE_MEMORY_TYPE;
  a->mem_map_full = device_map_full;
  a->mem_unmap_full = device_unmap_full;
  /* mem_copy: the GstAllocator default (CPU-maps the source -> a system-memory copy); no sub-memories */
  self->device = -1;
  GST_OBJECT_FLAG_SET (self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

/* the GPU ordinal an element runs on: its device-id property resolved like libvfhip does (-1: $VFHIP_DEVICE, else 0);
 * negative when there is no usable device */
gint
gst_vfhip_element_device (gpointer element)
{
  gint id = -1;
  if (element && g_object_class_find_property (G_OBJECT_GET_CLASS (element), "device-id"))
    g_object_get (element, "device-id", &id, NULL);
  return vfhip_device_init (id);
}

/* one allocator per GPU ordinal: streams sharded across the GPUs of a node by device-id keep their frames on their GPU */
GstAllocator *
gst_vfhip_device_allocator_get (gint device)
{
  static GMutex lock;
  static GstAllocator *alloc[64];
  GstAllocator *a;
  const gint slot = (device < 0 || device > 63) ? 0 : device;
  g_mutex_lock (&lock);
  if (!alloc[slot]) {
    alloc[slot] = g_object_new (gst_vfhip_device_allocator_get_type (), NULL);
    gst_object_ref_sink (alloc[slot]);
    ((GstVfHipDeviceAllocator *) alloc[slot])->device = slot;
  }
  a = gst_object_ref (alloc[slot]);
  g_mutex_unlock (&lock);
  return a;
}

/* GST_MAP_VFHIP when `buf` is one device memory on GPU `device` (its planes can be used in place); 0 otherwise — a device
 * memory that lives on ANOTHER GPU is then mapped through its host shadow like by any CPU element */
GstMapFlags
gst_vfhip_map_flag (GstBuffer * buf, gint device)
{
  GstMemory *mem;
  if (device < 0 || gst_buffer_n_memory (buf) != 1)
    return (GstMapFlags) 0;
  mem = gst_buffer_peek_memory (buf, 0);
  if (!gst_vfhip_is_device_memory (mem) || ((VfHipDeviceMemory *) mem)->device != device)
    return (GstMapFlags) 0;
  return GST_MAP_VFHIP;
}

gboolean
gst_vfhip_is_device_memory (GstMemory * mem)
{
  return mem && mem->allocator && G_OBJECT_TYPE (mem->allocator) == gst_vfhip_device_allocator_get_type ();
}

gboolean
gst_vfhip_caps_has_hip_feature (GstCaps * caps)
{
  GstCapsFeatures *f;
  if (!caps || gst_caps_is_empty (caps) || gst_caps_is_any (caps))
    return FALSE;
  f = gst_caps_get_features (caps, 0);
  return f && gst_caps_features_contains (f, GST_CAPS_FEATURE_MEMORY_HIP);
}

/* every structure of `caps` twice: once as memory:HIPMemory (first: preferred between two vfhip elements), once as
 * system memory — a vfhip element converts between the two for free (it uploads / downloads anyway) */
GstCaps *
gst_vfhip_caps_both_memories (GstCaps * caps)
{
  GstCaps *res = gst_caps_new_empty ();
  guint i, pass, n = gst_caps_get_size (caps);
  for (pass = 0; pass < 2; pass++)
    for (i = 0; i < n; i++) {
      GstStructure *st = gst_structure_copy (gst_caps_get_structure (caps, i));
      GstCapsFeatures *f = pass == 0 ? gst_caps_features_new (GST_CAPS_FEATURE_MEMORY_HIP, NULL) : gst_caps_features_new_empty ();
      GstCaps *one = gst_caps_new_full (st, NULL);
      gst_caps_set_features (one, 0, f);
      if (!gst_caps_is_subset (one, res))
        res = gst_caps_merge (res, one);
      else
        gst_caps_unref (one);
    }
  return res;
}

/* GstBaseTransform::transform_caps of the same-caps elements (GstVideoFilter subclasses): the other side may carry the
 * same video in either memory */
GstCaps *
gst_vfhip_filter_transform_caps (GstBaseTransform * trans, GstPadDirection direction, GstCaps * caps, GstCaps * filter)
{
  GstCaps *res = gst_vfhip_caps_both_memories (caps);
  (void) trans; (void) direction;
  if (filter) {
    GstCaps *tmp = gst_caps_intersect_full (filter, res, GST_CAPS_INTERSECT_FIRST);
    gst_caps_unref (res);
    res = tmp;
  }
  return res;
}

/* GstBaseTransform::transform of the GstVideoFilter subclasses: GstVideoFilter's own maps for CPU access, which would
 * pull a device buffer through the host shadow; this maps with GST_MAP_VFHIP and then calls transform_frame as usual */
GstFlowReturn
gst_vfhip_filter_transform (GstBaseTransform * trans, GstBuffer * inbuf, GstBuffer * outbuf)
{
  GstVideoFilter *filter = GST_VIDEO_FILTER_CAST (trans);
  GstVideoFilterClass *fclass = GST_VIDEO_FILTER_GET_CLASS (filter);
  GstVideoFrame in, out;
  GstFlowReturn res;
  GstVfHipPinStats *pin;
  gint dev;
  if (G_UNLIKELY (!filter->negotiated))
    return GST_FLOW_NOT_NEGOTIATED;
  if (!(pin = g_object_get_data (G_OBJECT (trans), "vfhip-pin-stats"))) {
    pin = g_new0 (GstVfHipPinStats, 1);
    g_object_set_data_full (G_OBJECT (trans), "vfhip-pin-stats", pin, g_free);
  }
  gst_vfhip_pin_foreign_memory (inbuf, pin);                 /* recurring pageable upstream memory: page-lock it in place */
  dev = gst_vfhip_element_device (trans);
  if (!gst_video_frame_map (&in, &filter->in_info, inbuf, (GstMapFlags) (GST_MAP_READ | gst_vfhip_map_flag (inbuf, dev) | GST_VIDEO_FRAME_MAP_FLAG_NO_REF)))
    return GST_FLOW_ERROR;
  if (!gst_video_frame_map (&out, &filter->out_info, outbuf, (GstMapFlags) (GST_MAP_WRITE | gst_vfhip_map_flag (outbuf, dev) | GST_VIDEO_FRAME_MAP_FLAG_NO_REF))) {
    gst_video_frame_unmap (&in);
    return GST_FLOW_ERROR;
  }
  res = fclass->transform_frame (filter, &in, &out);
  gst_video_frame_unmap (&out);
  gst_video_frame_unmap (&in);
  return res;
}
