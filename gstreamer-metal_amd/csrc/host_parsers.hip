// csrc/host_parsers.hip — the file parsers of libvfhip: a small PNG decoder (inflate comes from zlib), the .cube 3D-LUT
// parser and the PNG-LUT slicer.  Host code only and free of HIP headers (vfhip_host.h), so that the same file builds as
// plain C++ under AddressSanitizer / UBSan for the corrupt-input tests (tests/test_parsers_asan.py).
//
// The reference decodes overlay images and PNG LUTs with CoreGraphics / ImageIO (overlay/metaloverlayrenderer.m:166-245,
// videofilter/metalvideofilterrenderer.m:166-305), which do not exist here — SURVEY.md §8f item 4 lists "a PNG decoder"
// as the dependency of both.  Supported: non-interlaced PNG, bit depths 8 and 16 (high byte), colour types grey, RGB,
// palette (+ tRNS), grey+alpha, RGBA; all five scan-line filters.  Output: straight (non-premultiplied) RGBA8, row 0 first.
// Anything else (interlaced, depths 1/2/4, broken CRC-less streams that fail to inflate) is an error, never a guess.
#include "vfhip_host.h"
#include <zlib.h>
#include <cctype>
#include <cstdlib>
#include <new>
#include <strings.h>

using namespace vfhip;

namespace vfhip {

static uint32_t be32 (const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

static int paeth (int a, int b, int c)
{
  const int p = a + b - c, pa = abs (p - a), pb = abs (p - b), pc = abs (p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static int decode_png_impl (const char *path, std::vector<uint8_t> &rgba, int *width, int *height)
{
  FILE *f = fopen (path, "rb");
  if (!f) return set_error (VFHIP_ERR_INVALID, "cannot open %s", path);
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread (buf, 1, sizeof buf, f)) > 0) file.insert (file.end (), buf, buf + n);
  fclose (f);
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
  if (file.size () < 8 + 25 || memcmp (file.data (), sig, 8) != 0) return set_error (VFHIP_ERR_UNSUPPORTED, "%s is not a PNG file", path);
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  size_t pos = 8;
  bool end = false;
  while (!end && pos + 12 <= file.size ()) {
    const uint32_t len = be32 (&file[pos]);
    const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
    if ((size_t) len > file.size () - pos - 12) return set_error (VFHIP_ERR_INVALID, "%s: truncated PNG chunk", path);
    if (!memcmp (type, "IHDR", 4)) {
      if (len < 13) return set_error (VFHIP_ERR_INVALID, "%s: bad IHDR", path);
      w = (int) be32 (data); h = (int) be32 (data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (!memcmp (type, "PLTE", 4)) plte.assign (data, data + len);
    else if (!memcmp (type, "tRNS", 4)) trns.assign (data, data + len);
    else if (!memcmp (type, "IDAT", 4)) idat.insert (idat.end (), data, data + len);
    else if (!memcmp (type, "IEND", 4)) end = true;
    pos += 12 + (size_t) len;
  }
  if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || (size_t) w * (size_t) h > ((size_t) 64 << 20))
    return set_error (VFHIP_ERR_INVALID, "%s: bad PNG size %dx%d (at most 16384 per side, 64 Mpixel)", path, w, h);
  if (interlace) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: interlaced PNGs are not supported", path);
  if (depth != 8 && depth != 16) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: PNG bit depth %d is not supported (8 or 16)", path, depth);
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; if (depth != 8) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: palette PNG must be 8-bit", path); break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return set_error (VFHIP_ERR_UNSUPPORTED, "%s: PNG colour type %d", path, ctype);
  }
  const size_t bpp = (size_t) channels * depth / 8, stride = (size_t) w * bpp;
  std::vector<uint8_t> raw ((stride + 1) * (size_t) h);
  uLongf out_len = (uLongf) raw.size ();
  if (idat.empty () || uncompress (raw.data (), &out_len, idat.data (), (uLong) idat.size ()) != Z_OK || out_len != raw.size ())
    return set_error (VFHIP_ERR_INVALID, "%s: PNG image data does not inflate to %dx%d", path, w, h);
  // undo the scan-line filters in place
  std::vector<uint8_t> zero (stride, 0);
  for (int y = 0; y < h; y++) {
    uint8_t *row = &raw[(stride + 1) * (size_t) y];
    const int ft = row[0];
    uint8_t *cur = row + 1;
    const uint8_t *up = y ? &raw[(stride + 1) * (size_t) (y - 1) + 1] : zero.data ();
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
      int v = cur[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth (a, b, c); break;
        default: return set_error (VFHIP_ERR_INVALID, "%s: bad PNG filter type %d", path, ft);
      }
      cur[i] = (uint8_t) v;
    }
  }
  rgba.assign ((size_t) w * h * 4, 255);
  const int step = depth / 8;                       // 16-bit samples: the high byte
  for (int y = 0; y < h; y++) {
    const uint8_t *s = &raw[(stride + 1) * (size_t) y + 1];
    uint8_t *d = &rgba[(size_t) y * w * 4];
    for (int x = 0; x < w; x++, d += 4) {
      const uint8_t *p = s + (size_t) x * bpp;
      switch (ctype) {
        case 0: d[0] = d[1] = d[2] = p[0]; break;
        case 2: d[0] = p[0]; d[1] = p[step]; d[2] = p[2 * step]; break;
        case 3: {
          const size_t k = p[0];
          if (3 * k + 2 >= plte.size ()) return set_error (VFHIP_ERR_INVALID, "%s: palette index out of range", path);
          d[0] = plte[3 * k]; d[1] = plte[3 * k + 1]; d[2] = plte[3 * k + 2];
          if (k < trns.size ()) d[3] = trns[k];
          break;
        }
        case 4: d[0] = d[1] = d[2] = p[0]; d[3] = p[step]; break;
        default: d[0] = p[0]; d[1] = p[step]; d[2] = p[2 * step]; d[3] = p[3 * step]; break;
      }
    }
  }
  *width = w; *height = h;
  return VFHIP_OK;
}

int decode_png (const char *path, std::vector<uint8_t> &rgba, int *width, int *height)
{
  try { return decode_png_impl (path, rgba, width, height); }
  catch (const std::bad_alloc &) { return set_error (VFHIP_ERR_NOMEM, "%s: out of memory while decoding", path); }
}

// .cube parser: LUT_3D_SIZE 2..64, RGB triplets with R fastest; TITLE / DOMAIN_* / LUT_1D_SIZE lines are skipped
// (same acceptance rules as the reference's parse_cube_lut, videofilter/metalvideofilterrenderer.m:68-162).
int parse_cube_lut (const char *path, std::vector<float> &data, int *size_out)
{
  FILE *fp = fopen (path, "r");
  if (!fp) return set_error (VFHIP_ERR_IO, "cannot open LUT file %s", path);
  int size = 0;
  size_t count = 0, want = 0;
  char line[512];
  int rc = VFHIP_OK;
  try {
    while (fgets (line, sizeof (line), fp)) {
      const char *p = line;
      while (*p && isspace ((unsigned char) *p)) p++;
      if (*p == '#' || *p == '\0') continue;
      if (strncmp (p, "LUT_3D_SIZE", 11) == 0) {
        size = atoi (p + 11);
        if (size < 2 || size > 64) { rc = set_error (VFHIP_ERR_IO, "invalid LUT_3D_SIZE %d in %s", size, path); break; }
        want = (size_t) size * size * size;
        data.assign (want * 4, 1.0f);
        count = 0;
        continue;
      }
      if (strncmp (p, "TITLE", 5) == 0 || strncmp (p, "DOMAIN_MIN", 10) == 0 || strncmp (p, "DOMAIN_MAX", 10) == 0 || strncmp (p, "LUT_1D_SIZE", 11) == 0)
        continue;
      float r, g, b;
      if (size > 0 && count < want && sscanf (p, "%f %f %f", &r, &g, &b) == 3) {
        data[count * 4 + 0] = r; data[count * 4 + 1] = g; data[count * 4 + 2] = b; data[count * 4 + 3] = 1.0f;
        count++;
      }
    }
  } catch (const std::bad_alloc &) { rc = set_error (VFHIP_ERR_NOMEM, "%s: out of memory", path); }
  fclose (fp);
  if (rc) return rc;
  if (size == 0 || count != want) return set_error (VFHIP_ERR_IO, "incomplete .cube LUT %s: expected %zu entries, got %zu", path, want, count);
  *size_out = size;
  return VFHIP_OK;
}

// PNG LUT (reference parse_png_lut, videofilter/metalvideofilterrenderer.m:166-305): N^3 == width * height, slices of
// N x N pixels (r across, g down) laid out left to right, top to bottom, width / N per row; value / 255, alpha 1.
int parse_png_lut (const char *path, std::vector<float> &lut, int *size_out)
{
  std::vector<uint8_t> px;
  int w = 0, hh = 0;
  int rc = decode_png (path, px, &w, &hh);
  if (rc) return rc;
  int size = 0;
  for (int s = 2; s <= 64; s++) if ((long) s * s * s == (long) w * hh) { size = s; break; }      // the 3D texture limit (64) applies
  if (size == 0) return set_error (VFHIP_ERR_IO, "cannot determine the LUT size (2..64) from a %dx%d PNG (%s)", w, hh, path);
  const int per_row = w / size;
  if (per_row == 0 || ((size + per_row - 1) / per_row) * size > hh) return set_error (VFHIP_ERR_IO, "LUT PNG %s: %dx%d does not hold %d slices of %dx%d", path, w, hh, size, size, size);
  try { lut.assign ((size_t) size * size * size * 4, 1.0f); }
  catch (const std::bad_alloc &) { return set_error (VFHIP_ERR_NOMEM, "%s: out of memory", path); }
  for (int b = 0; b < size; b++)
    for (int g = 0; g < size; g++)
      for (int r = 0; r < size; r++) {
        const uint8_t *s8 = &px[((size_t) ((b / per_row) * size + g) * w + (size_t) (b % per_row) * size + r) * 4];
        float *d = &lut[(((size_t) b * size + g) * size + r) * 4];
        // CoreGraphics hands the reference PREMULTIPLIED bytes (kCGImageAlphaPremultipliedLast): identical for the opaque
        // PNGs LUTs are; for a translucent one its exact rounding is unpinned — (c * a + 127) / 255 here
        const unsigned a = s8[3];
        d[0] = (float) ((s8[0] * a + 127) / 255) / 255.0f; d[1] = (float) ((s8[1] * a + 127) / 255) / 255.0f; d[2] = (float) ((s8[2] * a + 127) / 255) / 255.0f;
      }
  *size_out = size;
  return VFHIP_OK;
}

}  // namespace vfhip

extern "C" int vfhip_image_decode_png (const char *path, uint8_t **rgba, int *width, int *height)
{
  using namespace vfhip;
  if (!path || !rgba || !width || !height) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::vector<uint8_t> px;
  int rc = decode_png (path, px, width, height);
  if (rc) return rc;
  *rgba = (uint8_t *) malloc (px.size ());
  if (!*rgba) return set_error (VFHIP_ERR_NOMEM, "out of memory");
  memcpy (*rgba, px.data (), px.size ());
  return VFHIP_OK;
}

extern "C" void vfhip_image_free (uint8_t *rgba) { free (rgba); }
