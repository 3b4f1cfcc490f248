/* oracle/metalref.h — TEST INFRASTRUCTURE ONLY.  See metalref.c. */
#ifndef METALREF_H
#define METALREF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MR_BGRA = 0, MR_RGBA = 1, MR_NV12 = 2, MR_I420 = 3, MR_UYVY = 4, MR_YUY2 = 5 };

typedef struct {
  uint8_t *p[3];
  int32_t s[3];
  int32_t w, h, fmt;
  int32_t m709;            /* 1: BT.709, 0: BT.601 (reference vf_metal_color_matrix_for_frame) */
} MrImg;

/* convertscale, reference float pipeline (metalconvertscale_shaders.h + metalconvertscalerenderer.m) */
int metalref_convertscale (const MrImg *in, const MrImg *out, int linear, int add_borders, uint32_t border_argb);

/* deinterlace (metaldeinterlace_shaders.h + metaldeinterlacerenderer.m); prev may be NULL (no history -> bob) */
enum { MR_DEINT_BOB = 0, MR_DEINT_WEAVE = 1, MR_DEINT_LINEAR = 2, MR_DEINT_GREEDYH = 3 };
int metalref_deinterlace (const MrImg *cur, const MrImg *prev, const MrImg *out, int method, int tff, float threshold);

/* videofilter (metalvideofilter_shaders.h + metalvideofilterrenderer.m) */
typedef struct {
  float brightness, contrast, saturation, hue, gamma, sharpness, sepia, noise, vignette;
  int32_t invert, chroma_key_enabled;
  float key_r, key_g, key_b, key_tolerance, key_smoothness;
  uint32_t frame_index;
} MrFilterParams;
int metalref_videofilter (const MrImg *in, const MrImg *out, const MrFilterParams *p, const float *lut_rgba, int lut_size);

/* compositor (metalcomprenderer.m) */
enum { MR_BLEND_SOURCE = 0, MR_BLEND_OVER = 1, MR_BLEND_ADD = 2 };
enum { MR_BG_CHECKER = 0, MR_BG_BLACK = 1, MR_BG_WHITE = 2, MR_BG_TRANSPARENT = 3 };
typedef struct {
  MrImg img;
  int32_t xpos, ypos, width, height;
  double alpha;
  int32_t blend;
} MrPad;
int metalref_compositor (const MrPad *pads, int n, int background, const MrImg *out);

/* transform (transform/metaltransform_shaders.h:40-120, metaltransformrenderer.m:44-104,265-293) */
int metalref_transform (const MrImg *in, const MrImg *out, int method, int crop_top, int crop_bottom, int crop_left, int crop_right);

/* overlay (overlay/metaloverlay_shaders.h:60-151, metaloverlayrenderer.m:247-300): `ov` = RGBA8 image as the reference's
 * texture holds it (already premultiplied by its decoder); ov == NULL -> plain copy through the 8-bit target */
int metalref_overlay (const MrImg *in, const MrImg *out, const MrImg *ov, float x, float y, float width, float height, float alpha);

#ifdef __cplusplus
}
#endif
#endif
