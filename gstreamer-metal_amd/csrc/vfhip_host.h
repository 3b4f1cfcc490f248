// csrc/vfhip_host.h — the part of libvfhip's internals that needs no HIP header: error reporting and the file parsers
// (host_parsers.hip).  Kept apart so that the parsers also compile as plain C++ (g++ -x c++ -fsanitize=address,undefined:
// tests/test_parsers_asan.py feeds them truncated and corrupt files on the CPU).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/vfhip.h"

namespace vfhip {

int set_error (int code, const char *fmt, ...) __attribute__ ((format (printf, 2, 3)));

// PNG -> straight RGBA8, row 0 first (zlib inflate; every colour type and bit depth, Adam7, at most 64 Mpixel)
int decode_png (const char *path, std::vector<uint8_t> &rgba, int *width, int *height);
// JPEG (sequential / progressive Huffman) -> RGBA8 (alpha 255), row 0 first (host_jpeg.hip); decode_image: PNG or JPEG by the file's first bytes
int decode_jpeg (const char *path, std::vector<uint8_t> &rgba, int *width, int *height);
int decode_image (const char *path, std::vector<uint8_t> &rgba, int *width, int *height);
// .cube 3D LUT -> size^3 RGBA float entries, R fastest, alpha 1 (reference parse_cube_lut, videofilter/metalvideofilterrenderer.m:68-162)
int parse_cube_lut (const char *path, std::vector<float> &rgba, int *size);
// PNG LUT (N x N slices, width / N per row) -> the same table (reference parse_png_lut, :166-305)
int parse_png_lut (const char *path, std::vector<float> &rgba, int *size);

}  // namespace vfhip
