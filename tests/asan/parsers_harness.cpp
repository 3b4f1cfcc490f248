// tests/asan/parsers_harness.cpp — TEST INFRASTRUCTURE: libvfhip's file parsers (csrc/host_parsers.hip, csrc/host_jpeg.hip, compiled as plain
// C++) under AddressSanitizer + UBSan.  Reads every file named on the command line through the parser its extension selects
// and prints one line per file; exits 0 whatever the parsers answer — only a sanitizer report (or a crash) fails the run.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <strings.h>
#include <vector>
#include "../../gstreamer-metal_amd/csrc/vfhip_host.h"

namespace vfhip {
static char g_err[512];
int set_error (int code, const char *fmt, ...)          // the library's own lives in vfhip_core.hip (HIP); same contract
{
  va_list ap; va_start (ap, fmt); vsnprintf (g_err, sizeof g_err, fmt, ap); va_end (ap);
  return code;
}
}

int main (int argc, char **argv)
{
  for (int i = 1; i < argc; i++) {
    const char *path = argv[i];
    const size_t n = strlen (path);
    vfhip::g_err[0] = 0;
    int rc, a = 0, b = 0;
    if (n >= 5 && !strcasecmp (path + n - 5, ".cube")) {
      std::vector<float> lut;
      rc = vfhip::parse_cube_lut (path, lut, &a);
      if (!rc && lut.size () != (size_t) a * a * a * 4) { printf ("%s: INCONSISTENT size %d table %zu\n", path, a, lut.size ()); return 3; }
    } else if (n >= 8 && !strcasecmp (path + n - 8, ".lut.png")) {
      std::vector<float> lut;
      rc = vfhip::parse_png_lut (path, lut, &a);
      if (!rc && lut.size () != (size_t) a * a * a * 4) { printf ("%s: INCONSISTENT size %d table %zu\n", path, a, lut.size ()); return 3; }
    } else if ((n >= 4 && !strcasecmp (path + n - 4, ".jpg")) || (n >= 4 && !strcasecmp (path + n - 4, ".img"))) {
      std::vector<uint8_t> px;                            // .jpg: the JPEG decoder; .img: the sniffing loader (PNG or JPEG by content)
      rc = !strcasecmp (path + n - 4, ".jpg") ? vfhip::decode_jpeg (path, px, &a, &b) : vfhip::decode_image (path, px, &a, &b);
      if (!rc && px.size () != (size_t) a * b * 4) { printf ("%s: INCONSISTENT %dx%d pixels %zu\n", path, a, b, px.size ()); return 3; }
    } else {
      std::vector<uint8_t> px;
      rc = vfhip::decode_png (path, px, &a, &b);
      if (!rc && px.size () != (size_t) a * b * 4) { printf ("%s: INCONSISTENT %dx%d pixels %zu\n", path, a, b, px.size ()); return 3; }
    }
    printf ("%s: rc %d %d %d %s\n", path, rc, a, b, vfhip::g_err);
  }
  return 0;
}
