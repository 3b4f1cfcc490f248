/* gst/gstvfhipallocator.c — pinned host memory for GstBuffers.
 *
 * The reference copies every plane twice per element per frame on the CPU (upload into a shared texture,
 * common/vfmetaltextureutil.m:108, and the read-back, common/vfmetalyuvoutput.m:138-176).  On a discrete GPU the
 * equivalent copies are PCIe DMA; when the GstBuffer memory is pinned (hipHostMalloc), libvfhip DMAs planes straight
 * from / into it (hipMemcpy2DAsync on the side streams) and the staging memcpy disappears.  The elements offer this
 * allocator upstream (propose_allocation) and use it for their own output buffers (decide_allocation).
 * When no HIP device is usable the allocator degrades to plain malloc'd memory (the elements then fail at the first
 * frame with a proper error, not here). */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif
#include <gst/base/gstbasetransform.h>
#include "gstvfhip.h"

#define GST_CAT_DEFAULT gst_vfhip_debug
#define VFHIP_PINNED_MEMORY_TYPE "VfHipPinnedMemory"

typedef struct
{
  GstMemory mem;
  gpointer data;
  gboolean pinned;
} VfHipPinnedMemory;

typedef struct
{
  GstAllocator parent;
} GstVfHipPinnedAllocator;
typedef struct
{
  GstAllocatorClass parent_class;
} GstVfHipPinnedAllocatorClass;

G_DEFINE_TYPE (GstVfHipPinnedAllocator, gst_vfhip_pinned_allocator, GST_TYPE_ALLOCATOR);

static GstMemory *
pinned_alloc (GstAllocator * allocator, gsize size, GstAllocationParams * params)
{
  VfHipPinnedMemory *m = g_slice_new0 (VfHipPinnedMemory);
  const gsize align = params->align | 63;                   /* at least 64-byte aligned rows for the DMA engine */
  const gsize maxsize = size + params->prefix + params->padding + align;
  gsize offset;
  m->data = vfhip_pinned_alloc (-1, maxsize);
  m->pinned = m->data != NULL;
  if (!m->data) {
    GST_INFO ("pinned allocation of %" G_GSIZE_FORMAT " bytes failed (%s): falling back to malloc", maxsize, vfhip_last_error_string ());
    m->data = g_malloc (maxsize);
  }
  offset = params->prefix;
  if (((guintptr) m->data + offset) & align)
    offset += (align + 1) - (((guintptr) m->data + offset) & align);
  gst_memory_init (GST_MEMORY_CAST (m), params->flags, allocator, NULL, maxsize, params->align, offset, size);
  return GST_MEMORY_CAST (m);
}

static void
pinned_free (GstAllocator * allocator, GstMemory * mem)
{
  VfHipPinnedMemory *m = (VfHipPinnedMemory *) mem;
  (void) allocator;
  if (m->pinned) vfhip_pinned_free (m->data);
  else g_free (m->data);
  g_slice_free (VfHipPinnedMemory, m);
}

static gpointer
pinned_map (GstMemory * mem, gsize maxsize, GstMapFlags flags)
{
  (void) maxsize; (void) flags;
  return ((VfHipPinnedMemory *) mem)->data;
}

static void
pinned_unmap (GstMemory * mem)
{
  (void) mem;
}

static void
gst_vfhip_pinned_allocator_class_init (GstVfHipPinnedAllocatorClass * klass)
{
  GST_ALLOCATOR_CLASS (klass)->alloc = pinned_alloc;
  GST_ALLOCATOR_CLASS (klass)->free = pinned_free;
}

static void
gst_vfhip_pinned_allocator_init (GstVfHipPinnedAllocator * self)
{
  GstAllocator *a = GST_ALLOCATOR_CAST (self);
  a->mem_type = VFHIP_PINNED_MEMORY_TYPE;
  a->mem_map = pinned_map;
  a->mem_unmap = pinned_unmap;
  /* mem_copy / mem_share / mem_is_span: the GstAllocator defaults (copy into system memory, no sub-memories) */
  GST_OBJECT_FLAG_SET (self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

GstAllocator *
gst_vfhip_pinned_allocator_get (void)
{
  static gsize once = 0;
  static GstAllocator *alloc = NULL;
  if (g_once_init_enter (&once)) {
    alloc = g_object_new (gst_vfhip_pinned_allocator_get_type (), NULL);
    gst_object_ref_sink (alloc);
    g_once_init_leave (&once, 1);
  }
  return gst_object_ref (alloc);
}

/* GstBaseTransform::propose_allocation: offer the pinned allocator (and video meta: libvfhip honours strides) upstream */
gboolean
gst_vfhip_propose_allocation (GstBaseTransform * trans, GstQuery * decide_query, GstQuery * query,
    gboolean (*parent) (GstBaseTransform *, GstQuery *, GstQuery *))
{
  if (!parent (trans, decide_query, query))
    return FALSE;
  if (decide_query != NULL) {                                 /* not in passthrough */
    GstCaps *caps = NULL;
    GstAllocator *a;
    GstAllocationParams params;
    gst_query_parse_allocation (query, &caps, NULL);
    /* memory:HIPMemory negotiated on our sink pad: upstream should hand us device buffers */
    a = gst_vfhip_caps_has_hip_feature (caps) ? gst_vfhip_device_allocator_get (gst_vfhip_element_device (trans)) : gst_vfhip_pinned_allocator_get ();
    gst_allocation_params_init (&params);
    params.align = 63;
    gst_query_add_allocation_param (query, a, &params);
    gst_query_add_allocation_meta (query, GST_VIDEO_META_API_TYPE, NULL);
    gst_object_unref (a);
  }
  return TRUE;
}

/* GstBaseTransform::decide_allocation: our own output buffers come from the pinned allocator, or — when the src caps
 * carry memory:HIPMemory — from the device allocator (any pool downstream proposed for system memory is dropped then) */
gboolean
gst_vfhip_decide_allocation (GstBaseTransform * trans, GstQuery * query, gboolean (*parent) (GstBaseTransform *, GstQuery *))
{
  GstCaps *caps = NULL;
  GstAllocator *a;
  GstAllocationParams params;
  gst_query_parse_allocation (query, &caps, NULL);
  if (gst_vfhip_caps_has_hip_feature (caps)) {
    a = gst_vfhip_device_allocator_get (gst_vfhip_element_device (trans));
    while (gst_query_get_n_allocation_pools (query) > 0)
      gst_query_remove_nth_allocation_pool (query, 0);
  } else
    a = gst_vfhip_pinned_allocator_get ();
  gst_allocation_params_init (&params);
  params.align = 63;
  if (gst_query_get_n_allocation_params (query) > 0)
    gst_query_set_nth_allocation_param (query, 0, a, &params);
  else
    gst_query_add_allocation_param (query, a, &params);
  gst_object_unref (a);
  return parent (trans, query);
}


/* ---- hipHostRegister for buffers we did not allocate -------------------------------------------------------------
 * Upstream elements that ignore the proposed allocator (their own pools, appsrc with application memory) hand us
 * pageable system memory, which libvfhip has to copy into its pinned staging buffer before the DMA.  Pools recycle
 * their memories, so a system-memory GstMemory is page-locked in place the first time it is seen (hipHostRegister)
 * and from then on its planes are DMA'd directly, like the pinned allocator's.  Lifetime is tied to the GstMemory:
 * a qdata destroy-notify unregisters the range before the memory is freed, so a recycled virtual address can never
 * alias a stale registration.  Registration costs ~1 ms per 12 MB, which only pays off when memories recur: after
 * VFHIP_REGISTER_PROBE registrations without a single re-use the element stops trying (fresh malloc per buffer). */
#define VFHIP_REGISTER_PROBE 12

static void
unregister_host_range (gpointer data)
{
  vfhip_host_unregister (data);
}

void
gst_vfhip_pin_foreign_memory (GstBuffer * buf, GstVfHipPinStats * stats)
{
  static GQuark quark = 0;
  guint i, n;
  if (!quark)
    quark = g_quark_from_static_string ("vfhip-host-registered");
  if (stats->disabled)
    return;
  n = gst_buffer_n_memory (buf);
  for (i = 0; i < n; i++) {
    GstMemory *mem = gst_buffer_peek_memory (buf, i);
    GstMapInfo info;
    if (!mem->allocator || !gst_memory_is_type (mem, GST_ALLOCATOR_SYSMEM) || mem->parent != NULL)
      continue;                                               /* ours (pinned / device), foreign types, or a sub-memory */
    if (gst_mini_object_get_qdata (GST_MINI_OBJECT_CAST (mem), quark)) {
      stats->reused++;
      continue;
    }
    if (stats->registered >= VFHIP_REGISTER_PROBE && stats->reused == 0) {
      GST_INFO ("upstream memory never recurs: giving up on hipHostRegister for this element");
      stats->disabled = TRUE;
      return;
    }
    if (mem->maxsize < 256 * 1024 || !gst_memory_map (mem, &info, GST_MAP_READ))
      continue;
    if (vfhip_host_register (info.data - mem->offset, mem->maxsize) == VFHIP_OK) {
      gst_mini_object_set_qdata (GST_MINI_OBJECT_CAST (mem), quark, info.data - mem->offset, unregister_host_range);
      stats->registered++;
      GST_INFO ("page-locked upstream memory %p (%" G_GSIZE_FORMAT " bytes) in place", info.data - mem->offset, mem->maxsize);
    } else {
      GST_DEBUG ("hipHostRegister failed: %s", vfhip_last_error_string ());
      stats->disabled = TRUE;
    }
    gst_memory_unmap (mem, &info);
  }
}
