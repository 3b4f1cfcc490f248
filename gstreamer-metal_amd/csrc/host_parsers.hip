// csrc/host_parsers.hip — the file parsers of libvfhip: a small PNG decoder (inflate comes from zlib), the .cube 3D-LUT
// parser and the PNG-LUT slicer.  Host code only and free of HIP headers (vfhip_host.h), so that the same file builds as
// plain C++ under AddressSanitizer / UBSan for the corrupt-input tests (tests/test_parsers_asan.py).
//
// The reference decodes overlay images and PNG LUTs with CoreGraphics / ImageIO (overlay/metaloverlayrenderer.m:166-245,
// videofilter/metalvideofilterrenderer.m:166-305), which do not exist here — SURVEY.md §8f item 4 lists "a PNG decoder"
// as the dependency of both.  Supported: every colour type at every bit depth the PNG specification allows (grey 1/2/4/8/16,
// RGB 8/16, palette 1/2/4/8, grey+alpha and RGBA 8/16; 16-bit samples keep their high byte, 1/2/4-bit grey scales to 0..255),
// tRNS in all three forms (palette alpha, grey key, RGB key), Adam7 interlacing, all five scan-line filters, IDAT split
// anywhere.  Output: straight (non-premultiplied) RGBA8, row 0 first.  Ancillary chunks (gAMA, sRGB, iCCP, ...) are ignored:
// sample values are taken as they are.  A stream that does not inflate to exactly the announced image is an error, never a guess.
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
  if (interlace > 1) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: PNG interlace method %d", path, interlace);
  int channels;
  bool depth_ok;
  switch (ctype) {
    case 0: channels = 1; depth_ok = depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16; break;
    case 2: channels = 3; depth_ok = depth == 8 || depth == 16; break;
    case 3: channels = 1; depth_ok = depth == 1 || depth == 2 || depth == 4 || depth == 8; break;
    case 4: channels = 2; depth_ok = depth == 8 || depth == 16; break;
    case 6: channels = 4; depth_ok = depth == 8 || depth == 16; break;
    default: return set_error (VFHIP_ERR_UNSUPPORTED, "%s: PNG colour type %d", path, ctype);
  }
  if (!depth_ok) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: PNG bit depth %d is not valid for colour type %d", path, depth, ctype);
  // the sub-images of the stream: one, or Adam7's seven (x0, y0, dx, dy); empty passes carry no bytes
  struct Pass { int x0, y0, dx, dy, pw, ph; size_t stride; };
  static const int adam7[7][4] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
  Pass pass[7];
  int n_pass = 0;
  const size_t bits = (size_t) channels * depth;
  size_t total = 0;
  for (int k = 0; k < (interlace ? 7 : 1); k++) {
    Pass q;
    if (interlace) { q.x0 = adam7[k][0]; q.y0 = adam7[k][1]; q.dx = adam7[k][2]; q.dy = adam7[k][3]; }
    else { q.x0 = q.y0 = 0; q.dx = q.dy = 1; }
    q.pw = (w - q.x0 + q.dx - 1) / q.dx; q.ph = (h - q.y0 + q.dy - 1) / q.dy;
    if (q.pw <= 0 || q.ph <= 0) continue;
    q.stride = ((size_t) q.pw * bits + 7) / 8;
    total += (q.stride + 1) * (size_t) q.ph;
    pass[n_pass++] = q;
  }
  const size_t bpp = bits >= 8 ? bits / 8 : 1;        // the filters' "corresponding byte" distance
  std::vector<uint8_t> raw (total);
  uLongf out_len = (uLongf) raw.size ();
  if (idat.empty () || uncompress (raw.data (), &out_len, idat.data (), (uLong) idat.size ()) != Z_OK || out_len != raw.size ())
    return set_error (VFHIP_ERR_INVALID, "%s: PNG image data does not inflate to %dx%d", path, w, h);
  if (ctype == 3 && plte.size () < 3) return set_error (VFHIP_ERR_INVALID, "%s: palette PNG without a PLTE chunk", path);
  // tRNS of the grey / truecolour types: ONE colour (16-bit big-endian per sample, compared at the file's depth) is transparent
  const bool key_grey = ctype == 0 && trns.size () >= 2, key_rgb = ctype == 2 && trns.size () >= 6;
  unsigned key[3] = { 0, 0, 0 };
  for (int k = 0; k < (key_rgb ? 3 : (key_grey ? 1 : 0)); k++) key[k] = ((unsigned) trns[2 * k] << 8) | trns[2 * k + 1];
  rgba.assign ((size_t) w * h * 4, 255);
  const int step = depth == 16 ? 2 : 1;               // 16-bit samples: the high byte is kept
  const unsigned grey_scale = depth < 8 ? 255u / ((1u << depth) - 1u) : 1u;     // 1, 2, 4 bit grey -> 8 bit: x255, x85, x17
  size_t at = 0;
  for (int k = 0; k < n_pass; k++) {
    const Pass &q = pass[k];
    // undo the scan-line filters of this sub-image in place
    std::vector<uint8_t> zero (q.stride, 0);
    for (int y = 0; y < q.ph; y++) {
      uint8_t *row = &raw[at + (q.stride + 1) * (size_t) y];
      const int ft = row[0];
      uint8_t *cur = row + 1;
      const uint8_t *up = y ? row - q.stride : zero.data ();
      for (size_t i = 0; i < q.stride; i++) {
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
    for (int y = 0; y < q.ph; y++) {
      const uint8_t *s = &raw[at + (q.stride + 1) * (size_t) y + 1];
      for (int x = 0; x < q.pw; x++) {
        uint8_t *d = &rgba[((size_t) (q.y0 + y * q.dy) * w + (size_t) (q.x0 + x * q.dx)) * 4];
        if (depth < 8) {
          // packed samples, leftmost in the high bits
          const unsigned v = (s[((size_t) x * depth) >> 3] >> (8 - depth - (((size_t) x * depth) & 7))) & ((1u << depth) - 1u);
          if (ctype == 3) {
            if (3 * (size_t) v + 2 >= plte.size ()) return set_error (VFHIP_ERR_INVALID, "%s: palette index out of range", path);
            d[0] = plte[3 * v]; d[1] = plte[3 * v + 1]; d[2] = plte[3 * v + 2];
            if (v < trns.size ()) d[3] = trns[v];
          } else {
            d[0] = d[1] = d[2] = (uint8_t) (v * grey_scale);
            if (key_grey && v == key[0]) d[3] = 0;
          }
          continue;
        }
        const uint8_t *p = s + (size_t) x * bpp;
        switch (ctype) {
          case 0:
            d[0] = d[1] = d[2] = p[0];
            if (key_grey && (depth == 16 ? (((unsigned) p[0] << 8) | p[1]) : p[0]) == key[0]) d[3] = 0;
            break;
          case 2:
            d[0] = p[0]; d[1] = p[step]; d[2] = p[2 * step];
            if (key_rgb) {
              bool hit = true;
              for (int c = 0; c < 3; c++) hit = hit && (depth == 16 ? (((unsigned) p[2 * c] << 8) | p[2 * c + 1]) : p[c]) == key[c];
              if (hit) d[3] = 0;
            }
            break;
          case 3: {
            const size_t v = p[0];
            if (3 * v + 2 >= plte.size ()) return set_error (VFHIP_ERR_INVALID, "%s: palette index out of range", path);
            d[0] = plte[3 * v]; d[1] = plte[3 * v + 1]; d[2] = plte[3 * v + 2];
            if (v < trns.size ()) d[3] = trns[v];
            break;
          }
          case 4: d[0] = d[1] = d[2] = p[0]; d[3] = p[step]; break;
          default: d[0] = p[0]; d[1] = p[step]; d[2] = p[2 * step]; d[3] = p[3 * step]; break;
        }
      }
    }
    at += (q.stride + 1) * (size_t) q.ph;
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
