/* tests/asan/oracle_harness.c — TEST INFRASTRUCTURE: the float oracle (oracle/metalref.c) under AddressSanitizer + UBSan on
 * small frames of awkward sizes (odd widths / heights, 1-pixel lines, up- and down-scales) in EXACTLY sized heap buffers:
 * an out-of-bounds access of the restatement itself would otherwise hide behind "the GPU agrees with the oracle".
 * (oracle/gst114.c runs its whole golden-vector suite under the sanitizers instead: tests/test_parsers_asan.py.) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/metalref.h"

static uint32_t rng = 0x9E3779B9u;
static uint8_t rnd8 (void) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return (uint8_t) rng; }

static size_t layout (int fmt, int w, int h, int off[3], int stride[3])      /* GstVideoInfo default layout (4-byte aligned strides) */
{
  const int r4w = (w + 3) & ~3, cw = (w + 1) / 2, ch = (h + 1) / 2;
  off[0] = off[1] = off[2] = 0; stride[0] = stride[1] = stride[2] = 0;
  switch (fmt) {
    case MR_BGRA: case MR_RGBA: stride[0] = 4 * w; return (size_t) 4 * w * h;
    case MR_NV12: stride[0] = r4w; stride[1] = r4w; off[1] = r4w * h; return (size_t) r4w * h + (size_t) r4w * ch;
    case MR_I420: { const int cs = (cw + 3) & ~3; stride[0] = r4w; stride[1] = stride[2] = cs; off[1] = r4w * h; off[2] = off[1] + cs * ch; return (size_t) off[2] + (size_t) cs * ch; }
    default: stride[0] = ((w + 1) / 2) * 4; return (size_t) stride[0] * h;
  }
}

static MrImg image (int fmt, int w, int h, uint8_t **mem)
{
  int off[3], st[3];
  const size_t n = layout (fmt, w, h, off, st);
  MrImg im; memset (&im, 0, sizeof im);
  *mem = malloc (n ? n : 1);                     /* exact size: the sanitizer sees one byte too many */
  for (size_t i = 0; i < n; i++) (*mem)[i] = rnd8 ();
  for (int k = 0; k < 3; k++) { im.p[k] = *mem + off[k]; im.s[k] = st[k]; }
  im.w = w; im.h = h; im.fmt = fmt; im.m709 = (w ^ h) & 1;
  return im;
}

int main (void)
{
  static const int sizes[][2] = { {1, 1}, {2, 2}, {3, 5}, {7, 3}, {16, 9}, {33, 17}, {64, 36} };
  static const int fmts[] = { MR_BGRA, MR_RGBA, MR_NV12, MR_I420, MR_UYVY, MR_YUY2 };
  int runs = 0;
  for (unsigned a = 0; a < sizeof sizes / sizeof sizes[0]; a++)
    for (unsigned b = 0; b < sizeof sizes / sizeof sizes[0]; b++)
      for (unsigned fi = 0; fi < 6; fi++)
        for (unsigned fo = 0; fo < 6; fo++) {
          uint8_t *mi, *mo;
          MrImg in = image (fmts[fi], sizes[a][0], sizes[a][1], &mi), out = image (fmts[fo], sizes[b][0], sizes[b][1], &mo);
          metalref_convertscale (&in, &out, (a + b) & 1, (a ^ b) & 1, 0x80102030u);
          free (mi); free (mo); runs++;
        }
  for (unsigned a = 0; a < sizeof sizes / sizeof sizes[0]; a++)
    for (unsigned fi = 0; fi < 4; fi++) {
      const int w = sizes[a][0], h = sizes[a][1];
      uint8_t *mc, *mp, *mo, *m2;
      MrImg cur = image (fmts[fi], w, h, &mc), prev = image (fmts[fi], w, h, &mp), out = image (fmts[fi], w, h, &mo);
      for (int method = 0; method < 4; method++) { metalref_deinterlace (&cur, method & 1 ? &prev : NULL, &out, method, a & 1, 0.1f); runs++; }
      MrFilterParams fp; memset (&fp, 0, sizeof fp);
      fp.contrast = 1.2f; fp.saturation = 0.8f; fp.gamma = 1.5f; fp.hue = 0.9f; fp.sharpness = a & 1 ? 0.5f : -0.5f; fp.sepia = 0.2f; fp.noise = 0.1f; fp.vignette = 0.3f;
      fp.invert = 1; fp.chroma_key_enabled = 1; fp.key_g = 1.0f; fp.key_tolerance = 0.3f; fp.key_smoothness = 0.1f;
      float lut[3 * 3 * 3 * 4];
      for (int i = 0; i < 108; i++) lut[i] = rnd8 () / 255.0f;
      metalref_videofilter (&cur, &out, &fp, lut, 3); runs++;
      MrImg big = image (MR_BGRA, 2 * w + 3, 2 * h + 1, &m2);
      MrPad pads[2]; memset (pads, 0, sizeof pads);
      pads[0].img = cur; pads[0].xpos = -1; pads[0].ypos = 1; pads[0].width = w + 2; pads[0].height = h + 1; pads[0].alpha = 0.7; pads[0].blend = MR_BLEND_OVER;
      pads[1].img = prev; pads[1].xpos = w; pads[1].ypos = h - 1; pads[1].width = w; pads[1].height = h; pads[1].alpha = 1.0; pads[1].blend = MR_BLEND_ADD;
      metalref_compositor (pads, 2, a & 3, &big); runs++;
      metalref_transform (&cur, &out, a & 7, 0, 0, 0, 0); runs++;
      free (mc); free (mp); free (mo); free (m2);
    }
  printf ("oracle harness: %d runs, no sanitizer report\n", runs);
  return 0;
}
