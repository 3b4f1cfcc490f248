// csrc/compositor.hip — vfhip_compositor_* : N-input alpha / z-order compositor.
// Mirrors MetalCompositorRenderer (reference compositor/metalcomprenderer.{h,m}) and restates compositorVertex /
// compositorFragment{,NV12,I420} / checkerFragment (:39-122) and the fixed-function SOURCE / OVER / ADD blend
// states (:199-239) in `metal` numerics (float on unorm8, premultiplied source, the 8-bit target requantised
// after every layer).
//
// The reference clears the target, then draws N quads with ROP read-modify-write, then runs RGBA->YUV and reads
// back (metalcomprenderer.m:356-542): every covered output pixel is read and written once per layer.  Here one
// kernel walks the layers in z-order per output pixel with the running colour in a register, so each output
// pixel is written once and each covered input texel is read once; the per-layer 8-bit requantisation of the
// target is kept (it is part of the reference's arithmetic), it just happens in registers.
#include "vfhip_internal.h"
#include "metal_common.h"
#include <cstdlib>
#include <algorithm>

using namespace vfhip;

namespace vfhip {

constexpr int COMP_MAX_LAYERS = 16;     // per launch; more pads chain through an RGBA8 scratch target
constexpr int COMP_BG_IN_PLACE = -2;
constexpr int COMP_MAX_COVER = 8;

struct CompLayer {
  metal::Img img;
  int xpos, ypos, width, height;
  float alpha;
  int blend;
  size_t pitch;                         // batch: this pad's frame z at base + z * pitch
};
struct CompParams {
  CompLayer layer[COMP_MAX_LAYERS];
  int n;
  int background;                       // VfHipBackground; -1: start from `prev` (logical RGBA8 of an earlier pass); COMP_BG_IN_PLACE:
                                        // start from what `out` (RGBA / BGRA) already holds — a later draw over part of the frame
  const uint32_t *prev; int prev_stride;
  int bx0, by0;                         // the launch's first lane block (region launches; 0 for a full frame)
  int n_cover;                          // k_compositor_quads / k_compositor_420 drawing a frame's FIRST run: rectangles (x0, y0, x1, y1) that a later
  int cover[COMP_MAX_COVER][4];         // opaque pad overwrites completely — a wave whose strip lies inside one draws nothing
  metal::OutImg out;
  uint32_t *scratch; int scratch_stride;   // != nullptr: write logical RGBA8 here instead of `out`
  size_t out_pitch;                     // batch: output frame z at base + z * out_pitch (single-pass launches only)
};

using metal::F4;

// TARGET ORDER: the running colour q of a pixel is kept in the byte order of the output when that is RGBA or BGRA (logical RGBA
// for the planar outputs), so it is loaded from and stored to such an output as it is; the blend treats R, G and B alike, the
// backgrounds have R = B, and only a layer's own colour is brought into target order: a texel by the byte permute that unpacks
// it anyway, a sampled colour by exchanging two registers.
__device__ __forceinline__ uint32_t comp_swap_rb (uint32_t v, bool swap) { return __builtin_amdgcn_perm (0u, v, swap ? 0x03000102u : 0x03020100u); }
__device__ __forceinline__ F4 comp_order (F4 c, bool bgra_target) { if (bgra_target) { const float t = c.r; c.r = c.b; c.b = t; } return c; }

__device__ __forceinline__ uint32_t comp_background (const CompParams &p, const metal::OutImg &o, int x, int y)
{
  if (p.background == COMP_BG_IN_PLACE) return *reinterpret_cast<const uint32_t *> (o.p[0] + (size_t) y * o.s[0] + 4 * x);       // already in target order
  if (p.background < 0) return p.prev[(size_t) y * p.prev_stride + x];
  if (p.background == VFHIP_BG_BLACK) return 0xff000000u;
  if (p.background == VFHIP_BG_WHITE) return 0xffffffffu;
  if (p.background == VFHIP_BG_TRANSPARENT) return 0u;
  // checker: pos = int2 (texcoord * size) with texcoord = (x + .5) / size, 8x8 cells, grey 0.75 / 0.5 (metalcomprenderer.m:113-121).
  // int ((x + .5) / w * w) == x for every frame size the API admits (two roundings of 2^-24 relative on a value < 32768.5 cannot
  // reach the .5 distance to the next integer), so the cell parity comes from the integer coordinates; unorm8 (.75) = 191, (.5) = 128.
  return ((x >> 3) + (y >> 3)) & 1 ? 0xffbfbfbfu : 0xff808080u;
}

// the layer's colour at output pixel (x, y); the caller has checked coverage
__device__ __forceinline__ F4 comp_sample (const CompLayer &L, const metal::Img &im, int x, int y)
{
  if (L.width == L.img.w && L.height == L.img.h) {
    // unscaled pad: texel centres are sampled, the linear sampler returns the exact texel (SURVEY.md Appendix B
    // item 2); 4:2:0 chroma still interpolates at its .25/.75 phases.  One dword load instead of 16 byte taps.
    return metal::fetch_1to1 (im, x - L.xpos, y - L.ypos, true);
  }
  const float tu = (((float) x + 0.5f) - (float) L.xpos) / (float) L.width;
  const float tv = (((float) y + 0.5f) - (float) L.ypos) / (float) L.height;
  return metal::sample_rgba (im, tu, tv, true);
}

__device__ __forceinline__ F4 comp_texel (uint32_t t, bool in_order)
{
  // one byte permute (wave-uniform selector) puts a texel of the other byte order into target order, instead of two copies of the conversions
  return metal::unpack_rgba8 (comp_swap_rb (t, !in_order));
}

// colour s of one layer drawn over the target value d (the 8-bit target read back as floats).
// One expression serves the three operators: out = d * k + s with k = 0 (source), 1 (add), 1 - s.a (over).  fmaf (d, 0, s) = s and
// fmaf (d, 1, s) = s + d exactly for the finite non-negative values here, so this is compositorFragment's arithmetic
// (oracle/metalref.c comp_blend) bit for bit, in a third of the code.
__device__ __forceinline__ uint32_t comp_blend_f (const CompLayer &L, F4 s, const F4 &d)
{
  s.a *= L.alpha; s.r *= s.a; s.g *= s.a; s.b *= s.a;            // premultiply (compositorFragment, :58-59)
  const float k = L.blend == VFHIP_BLEND_SOURCE ? 0.0f : (L.blend == VFHIP_BLEND_ADD ? 1.0f : 1.0f - s.a);
  F4 o;
  o.r = fmaf (d.r, k, s.r); o.g = fmaf (d.g, k, s.g); o.b = fmaf (d.b, k, s.b); o.a = fmaf (d.a, k, s.a);
  return metal::quant_rgba8 (o);
}
// ... over the 8-bit target value q.  `flat`: wave-uniform, the target still holds the uniform background colour `bgc` under every
// lane (the first layer a wave draws over a black / white / transparent background): the target's read-back is then a scalar,
// not four conversions per pixel
__device__ __forceinline__ uint32_t comp_blend (const CompLayer &L, const F4 &s, uint32_t q, bool flat, const F4 &bgc)
{
  return flat ? comp_blend_f (L, s, bgc) : comp_blend_f (L, s, metal::unpack_rgba8 (q));
}

typedef uint4 __attribute__ ((aligned (4))) uint4_a4;

// the wave's strip [wx0, wx1) x [wy0, wy1) lies inside a rectangle that a later opaque pad overwrites (wave-uniform)
__device__ __forceinline__ bool comp_covered (const CompParams &p, int wx0, int wx1, int wy0, int wy1)
{
  bool c = false;
#pragma unroll
  for (int k = 0; k < COMP_MAX_COVER; k++)
    c = c || (k < p.n_cover && wx0 >= p.cover[k][0] && wy0 >= p.cover[k][1] && wx1 <= p.cover[k][2] && wy1 <= p.cover[k][3]);
  return c;
}

// k_compositor: the general kernel (any mix of scaled and unscaled pads).
// Workgroup = 64 x 4 lanes; one lane = a 4 x 2 block of output pixels (two of the store epilogue's 2x2 blocks), so a wave
// covers a 256 x 2 pixel strip and a lane carries eight independent blend chains (the first versions — one pixel, then
// a 2x2 block per lane — were latency-bound: ~40 % VALU-busy with one dependent load -> blend chain per layer).
// Layers are the OUTER loop: a layer's parameters are fetched once per wave (scalar loads) and a layer that misses the
// wave's strip is skipped by a wave-uniform branch.  An unscaled RGBA / BGRA pad that covers all four columns of a lane
// is fetched as ONE 16-byte load per row; RGBA / BGRA outputs leave as one 16-byte non-temporal store per row.
__global__ __launch_bounds__ (256) void k_compositor (const CompParams p)
{
  const int bx = p.bx0 + blockIdx.x * 64 + threadIdx.x;                                       // 4-pixel column group
  const int by = __builtin_amdgcn_readfirstlane ((int) (p.by0 + blockIdx.y * 4 + threadIdx.y));      // one row of lanes = one wave
  if (2 * by >= p.out.h) return;
  const bool live = 4 * bx < p.out.w;
  int xs[4];
#pragma unroll
  for (int i = 0; i < 4; i++) xs[i] = min (4 * bx + i, p.out.w - 1);
  const int y0 = 2 * by, y1 = min (2 * by + 1, p.out.h - 1);
  const unsigned z = blockIdx.z;
  const metal::OutImg o = metal::out_at (p.out, z * p.out_pitch);
  const bool bgra_out = o.fmt == VFHIP_FORMAT_BGRA;                   // q is kept in TARGET ORDER
  uint32_t q[2][4];
#pragma unroll
  for (int i = 0; i < 4; i++) { q[0][i] = comp_background (p, o, xs[i], y0); q[1][i] = comp_background (p, o, xs[i], y1); }
  const int wx0 = 4 * (p.bx0 + (int) blockIdx.x * 64), wx1 = wx0 + 256;
  // which layers touch this wave's 256 x 2 strip?  All sixteen rectangles are tested up front: independent scalar loads that
  // pipeline into one memory latency (testing inside the layer loop chained one dependent scalar-load latency per layer, five
  // per wave on BASELINE configs[3], most of them for layers the wave never draws)
  uint32_t hit = 0;
#pragma unroll
  for (int k = 0; k < COMP_MAX_LAYERS; k++) {
    const CompLayer &L = p.layer[k];
    const bool miss = k >= p.n || wx1 <= L.xpos || wx0 >= L.xpos + L.width || y1 < L.ypos || y0 >= L.ypos + L.height;
    hit |= miss ? 0u : 1u << k;
  }
  hit = (uint32_t) __builtin_amdgcn_readfirstlane ((int) hit);                                 // wave-uniform by construction
  // a uniform background reads back as one scalar colour until the first layer has been drawn
  bool flat = p.background == VFHIP_BG_BLACK || p.background == VFHIP_BG_WHITE || p.background == VFHIP_BG_TRANSPARENT;
  F4 bgc;
  bgc.r = bgc.g = bgc.b = p.background == VFHIP_BG_WHITE ? 1.0f : 0.0f; bgc.a = p.background == VFHIP_BG_TRANSPARENT ? 0.0f : 1.0f;     // un8 (0) / un8 (255)
  for (; hit; flat = false) {
    const int k = __builtin_ctz (hit);
    hit &= hit - 1;
    const CompLayer &L = p.layer[k];
    // a pixel is covered when its centre lies inside the quad [xpos, xpos+width) x [ypos, ypos+height)
    const int lx1 = L.xpos + L.width, ly1 = L.ypos + L.height;
    const metal::Img im = metal::img_at (L.img, z * L.pitch);
    bool cx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cx[i] = xs[i] >= L.xpos && xs[i] < lx1;
    const bool cy0 = y0 >= L.ypos && y0 < ly1, cy1 = y1 >= L.ypos && y1 < ly1;
    const bool rgba_in = (im.fmt == VFHIP_FORMAT_RGBA) != bgra_out;        // the pad's texels are in target order
    const bool quad = cx[0] && cx[3] && xs[3] == xs[0] + 3 && (im.fmt == VFHIP_FORMAT_RGBA || im.fmt == VFHIP_FORMAT_BGRA) && L.width == L.img.w && L.height == L.img.h;
    if (quad) {
      uint4 t[2];
#pragma unroll
      for (int r = 0; r < 2; r++)                       // both rows' loads first, then the eight blends
        if (r ? cy1 : cy0) t[r] = *reinterpret_cast<const uint4_a4 *> (im.p[0] + (size_t) ((r ? y1 : y0) - L.ypos) * im.s[0] + 4 * (xs[0] - L.xpos));
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (!(r ? cy1 : cy0)) continue;
        q[r][0] = comp_blend (L, comp_texel (t[r].x, rgba_in), q[r][0], flat, bgc);
        q[r][1] = comp_blend (L, comp_texel (t[r].y, rgba_in), q[r][1], flat, bgc);
        q[r][2] = comp_blend (L, comp_texel (t[r].z, rgba_in), q[r][2], flat, bgc);
        q[r][3] = comp_blend (L, comp_texel (t[r].w, rgba_in), q[r][3], flat, bgc);
      }
    } else if (cx[0] && cx[3] && xs[3] == xs[0] + 3 && cy0 && cy1 && y1 == y0 + 1 && (im.fmt == VFHIP_FORMAT_NV12 || im.fmt == VFHIP_FORMAT_I420) &&
               L.width == L.img.w && L.height == L.img.h && !(((xs[0] - L.xpos) | (y0 - L.ypos)) & 1)) {
      // unscaled 4:2:0 pad whose chroma grid is aligned with the lane's 4 x 2 block: the 8 pixels share 3 chroma rows x 4 chroma
      // columns.  fetch_1to1's bilinear chroma (phases .25 / .75) from 12 (U, V) fetches + 2 luma dwords instead of 72 byte loads,
      // the horizontal interpolation of the middle chroma row shared by both pixel rows.  Same operations per value as
      // metal::fetch_1to1 / plane_taps, so the result is bit-identical to the general path.
      const int px = xs[0] - L.xpos, py = y0 - L.ypos, j = px >> 1, m = py >> 1;
      const int cw = (im.w + 1) >> 1, chh = (im.h + 1) >> 1;
      typedef uint32_t __attribute__ ((aligned (1))) u32_any;
      typedef uint16_t __attribute__ ((aligned (1))) u16_any;
      uint32_t Y[2];
      Y[0] = *reinterpret_cast<const u32_any *> (im.p[0] + (size_t) py * im.s[0] + px);
      Y[1] = *reinterpret_cast<const u32_any *> (im.p[0] + (size_t) (py + 1) * im.s[0] + px);
      float cu[3][4], cv[3][4];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int row = metal::iclamp (m - 1 + r, 0, chh - 1);
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int col = metal::iclamp (j - 1 + c, 0, cw - 1);
          if (im.fmt == VFHIP_FORMAT_NV12) {
            const uint32_t uv = *reinterpret_cast<const u16_any *> (im.p[1] + (size_t) row * im.s[1] + 2 * col);
            cu[r][c] = metal::un8 (uv & 0xffu); cv[r][c] = metal::un8 (uv >> 8);
          } else {
            cu[r][c] = metal::un8 (im.p[1][(size_t) row * im.s[1] + col]); cv[r][c] = metal::un8 (im.p[2][(size_t) row * im.s[2] + col]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        // pixel px + i samples chroma at 0.5 (px + i) - 0.25: taps (j - 1 + (i + 1) / 2 ...), weight .75 for even i, .25 for odd i
        const int a = (i + 1) >> 1;                          // index of the first tap in cu[][0..3]: i = 0 -> 0, 1 -> 1, 2 -> 1, 3 -> 2
        const float fx = (i & 1) ? 0.25f : 0.75f;
        float hu[3], hv[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { hu[r] = metal::lerp2 (cu[r][a], cu[r][a + 1], fx); hv[r] = metal::lerp2 (cv[r][a], cv[r][a + 1], fx); }
#pragma unroll
        for (int r = 0; r < 2; r++) {
          const float fy = r ? 0.25f : 0.75f;
          const float cb = metal::lerp2 (hu[r], hu[r + 1], fy), cr = metal::lerp2 (hv[r], hv[r + 1], fy);
          const F4 c = comp_order (metal::yuv_to_rgb (metal::un8 ((Y[r] >> (8 * i)) & 0xffu), cb, cr, im.m709), bgra_out);
          q[r][i] = comp_blend (L, c, q[r][i], flat, bgc);
        }
      }
    } else {
      // general path (scaled pads, 4:2:0 pads off the chroma grid, quad edges): ONE instance of the sampler, the eight pixels take turns
      // (unrolled, the inlined samplers cost > 128 VGPRs)
#pragma unroll 1
      for (int i = 0; i < 8; i++) {
        const int c = i & 3, r = i >> 2;
        const bool cxi = c == 0 ? cx[0] : (c == 1 ? cx[1] : (c == 2 ? cx[2] : cx[3]));
        if (!(cxi && (r ? cy1 : cy0))) continue;
        const int xi = c == 0 ? xs[0] : (c == 1 ? xs[1] : (c == 2 ? xs[2] : xs[3]));
        uint32_t cur = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) cur = i == j ? q[j >> 2][j & 3] : cur;
        const uint32_t v = comp_blend (L, comp_order (comp_sample (L, im, xi, r ? y1 : y0), bgra_out), cur, flat, bgc);
#pragma unroll
        for (int j = 0; j < 8; j++) q[j >> 2][j & 3] = i == j ? v : q[j >> 2][j & 3];
      }
    }
  }
  if (!live) return;
  if (p.scratch) {
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
      for (int dx = 0; dx < 4; dx++)
        if (4 * bx + dx < p.out.w && 2 * by + dy < p.out.h) p.scratch[(size_t) (2 * by + dy) * p.scratch_stride + 4 * bx + dx] = q[dy][dx];
    return;
  }
  if ((o.fmt == VFHIP_FORMAT_BGRA || o.fmt == VFHIP_FORMAT_RGBA) && 4 * bx + 3 < o.w && !(((uintptr_t) o.p[0] | (uintptr_t) o.s[0]) & 15)) {
    typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
#pragma unroll
    for (int dy = 0; dy < 2; dy++) {
      if (2 * by + dy >= o.h) break;
      const v4u v = { q[dy][0], q[dy][1], q[dy][2], q[dy][3] };                              // target order: stored as it is
      __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (o.p[0] + (size_t) (2 * by + dy) * o.s[0]) + bx);
    }
    return;
  }
  // store_block takes logical RGBA (an unaligned or odd-width BGRA output comes this way too)
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 4; dx++) q[dy][dx] = comp_swap_rb (q[dy][dx], bgra_out);
  const uint32_t qa[2][2] = { { q[0][0], q[0][1] }, { q[1][0], q[1][1] } };
  metal::store_block (o, 2 * bx, by, qa);
  if (4 * bx + 2 < o.w) {
    const uint32_t qb[2][2] = { { q[0][2], q[0][3] }, { q[1][2], q[1][3] } };
    metal::store_block (o, 2 * bx + 1, by, qb);
  }
}


// k_compositor_unscaled: every pad of the launch is drawn at its own size (the common case: BASELINE configs[3], picture-in-picture,
// mosaics of equal tiles), so no layer needs the scaling sampler and its registers.
// Workgroup = 64 x 4 lanes; one lane = a 4 x 4 block of output pixels (COMP_PAIRS = 2 of the store epilogue's row pairs), so a
// wave covers a 256 x 4 pixel strip and a lane carries sixteen independent blend chains.  History: one pixel, then a 2x2 block
// per lane were latency-bound (~40 % VALU-busy, one dependent load -> blend chain per layer); 4 x 2 still had only ~40 KB of loads
// in flight per CU and ran 4 quadrant copies at 4.2 TB/s against the 6.2 TB/s of a plain copy (profiles/r02h_c4_variants.jsonl);
// 4 x 4 doubles the bytes in flight per wave and halves the per-wave set-up (rectangle tests, layer parameters).
// Layers are the OUTER loop: a layer's parameters are fetched once per wave (scalar loads) and a layer that misses the wave's
// strip is never visited.  An unscaled RGBA / BGRA pad that covers all four columns of a lane is fetched as ONE 16-byte load
// per row, all rows first; RGBA / BGRA outputs leave as one 16-byte non-temporal store per row.
constexpr int COMP_PAIRS = 2, COMP_ROWS = 2 * COMP_PAIRS;

__global__ __launch_bounds__ (256) void k_compositor_unscaled (const CompParams p)
{
  const int bx = p.bx0 + blockIdx.x * 64 + threadIdx.x;                                       // 4-pixel column group
  const int by = __builtin_amdgcn_readfirstlane ((int) (p.by0 + blockIdx.y * 4 + threadIdx.y));      // one row of lanes = one wave
  if (COMP_ROWS * by >= p.out.h) return;
  const bool live = 4 * bx < p.out.w;
  int xs[4], ys[COMP_ROWS];
#pragma unroll
  for (int i = 0; i < 4; i++) xs[i] = min (4 * bx + i, p.out.w - 1);
#pragma unroll
  for (int r = 0; r < COMP_ROWS; r++) ys[r] = min (COMP_ROWS * by + r, p.out.h - 1);          // edge-clamped duplicates below the frame
  const unsigned z = blockIdx.z;
  const metal::OutImg o = metal::out_at (p.out, z * p.out_pitch);
  const bool bgra_out = o.fmt == VFHIP_FORMAT_BGRA;                   // q is kept in TARGET ORDER
  uint32_t q[COMP_ROWS][4];
#pragma unroll
  for (int r = 0; r < COMP_ROWS; r++)
#pragma unroll
    for (int i = 0; i < 4; i++) q[r][i] = comp_background (p, o, xs[i], ys[r]);
  const int wx0 = 4 * (p.bx0 + (int) blockIdx.x * 64), wx1 = wx0 + 256, wy0 = ys[0], wy1 = ys[COMP_ROWS - 1];
  // which layers touch this wave's strip?  All sixteen rectangles are tested up front: independent scalar loads that pipeline
  // into one memory latency (testing inside the layer loop chained one dependent scalar-load latency per layer)
  uint32_t hit = 0;
#pragma unroll
  for (int k = 0; k < COMP_MAX_LAYERS; k++) {
    const CompLayer &L = p.layer[k];
    const bool miss = k >= p.n || wx1 <= L.xpos || wx0 >= L.xpos + L.width || wy1 < L.ypos || wy0 >= L.ypos + L.height;
    hit |= miss ? 0u : 1u << k;
  }
  hit = (uint32_t) __builtin_amdgcn_readfirstlane ((int) hit);                                 // wave-uniform by construction
  // a uniform background reads back as one scalar colour until the first layer has been drawn
  bool flat = p.background == VFHIP_BG_BLACK || p.background == VFHIP_BG_WHITE || p.background == VFHIP_BG_TRANSPARENT;
  F4 bgc;
  bgc.r = bgc.g = bgc.b = p.background == VFHIP_BG_WHITE ? 1.0f : 0.0f; bgc.a = p.background == VFHIP_BG_TRANSPARENT ? 0.0f : 1.0f;     // un8 (0) / un8 (255)
  for (; hit; flat = false) {
    const int k = __builtin_ctz (hit);
    hit &= hit - 1;
    const CompLayer &L = p.layer[k];
    // a pixel is covered when its centre lies inside the quad [xpos, xpos+width) x [ypos, ypos+height)
    const int lx1 = L.xpos + L.width, ly1 = L.ypos + L.height;
    const metal::Img im = metal::img_at (L.img, z * L.pitch);
    bool cx[4], cy[COMP_ROWS];
#pragma unroll
    for (int i = 0; i < 4; i++) cx[i] = xs[i] >= L.xpos && xs[i] < lx1;
#pragma unroll
    for (int r = 0; r < COMP_ROWS; r++) cy[r] = ys[r] >= L.ypos && ys[r] < ly1;
    const bool rgba_in = (im.fmt == VFHIP_FORMAT_RGBA) != bgra_out;        // the pad's texels are in target order
    const bool unscaled = true, full_x = cx[0] && cx[3] && xs[3] == xs[0] + 3;       // the host checked every layer (comp_launch)
    if (full_x && unscaled && (im.fmt == VFHIP_FORMAT_RGBA || im.fmt == VFHIP_FORMAT_BGRA)) {
      uint4 t[COMP_ROWS];
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++)                 // every row's load first, then the blends
        if (cy[r]) t[r] = *reinterpret_cast<const uint4_a4 *> (im.p[0] + (size_t) (ys[r] - L.ypos) * im.s[0] + 4 * (xs[0] - L.xpos));
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++) {
        if (!cy[r]) continue;
        q[r][0] = comp_blend (L, comp_texel (t[r].x, rgba_in), q[r][0], flat, bgc);
        q[r][1] = comp_blend (L, comp_texel (t[r].y, rgba_in), q[r][1], flat, bgc);
        q[r][2] = comp_blend (L, comp_texel (t[r].z, rgba_in), q[r][2], flat, bgc);
        q[r][3] = comp_blend (L, comp_texel (t[r].w, rgba_in), q[r][3], flat, bgc);
      }
      continue;
    }
    const bool yuv420 = im.fmt == VFHIP_FORMAT_NV12 || im.fmt == VFHIP_FORMAT_I420;
    if (full_x && yuv420 && im.w >= 7 && cy[0] && cy[COMP_ROWS - 1] && ys[COMP_ROWS - 1] == ys[0] + COMP_ROWS - 1 && !(((xs[0] - L.xpos) | (ys[0] - L.ypos)) & 1)) {
      // unscaled 4:2:0 pad whose chroma grid is aligned with the lane's 4 x 4 block: the 16 pixels share 4 chroma rows x 4 chroma
      // columns.  fetch_1to1's bilinear chroma (phases .25 / .75) from 4 chroma fetches of 8 (NV12) or 2 x 4 (I420) bytes + 4 luma
      // dwords, ALL issued before the first conversion, instead of 144 byte loads; the chroma rows then stream through the
      // horizontal interpolation one at a time (two rows of eight floats live) and each pair of consecutive rows yields the
      // pixel rows between them.  Same operations per value as metal::fetch_1to1 / plane_taps -> bit-identical to the general path.
      const int px = xs[0] - L.xpos, py = ys[0] - L.ypos, j = px >> 1, m = py >> 1;
      const int cw = (im.w + 1) >> 1, chh = (im.h + 1) >> 1;
      typedef uint32_t __attribute__ ((aligned (1))) u32_any;
      typedef uint2 __attribute__ ((aligned (1))) u64_any;
      uint32_t Y[COMP_ROWS], Ud[COMP_ROWS], Vd[COMP_ROWS];                 // Ud / Vd: chroma columns j - 1 .. j + 2, one byte each
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++) Y[r] = *reinterpret_cast<const u32_any *> (im.p[0] + (size_t) (py + r) * im.s[0] + px);
      const bool nv12 = im.fmt == VFHIP_FORMAT_NV12;
      // columns j - 1 .. j + 2 clamp to the plane only in a pad's first lane (j = 0: 0 0 1 2) and last lane (j + 2 = cw: j-1 j j+1 j+1):
      // fetch the four columns from `start` and let the byte permute that de-interleaves NV12 duplicate the edge column
      const int start = metal::iclamp (j - 1, 0, cw - 4);
      const uint32_t selu = j < 1 ? 0x04020000u : (j + 2 >= cw ? 0x06060402u : 0x06040200u), selv = selu + 0x01010101u;
      const uint32_t selp = j < 1 ? 0x02010000u : (j + 2 >= cw ? 0x03030201u : 0x03020100u);
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++) {
        const size_t row = (size_t) metal::iclamp (m - 1 + r, 0, chh - 1);
        if (nv12) {
          const uint2 v = *reinterpret_cast<const u64_any *> (im.p[1] + row * im.s[1] + 2 * start);
          Ud[r] = __builtin_amdgcn_perm (v.y, v.x, selu); Vd[r] = __builtin_amdgcn_perm (v.y, v.x, selv);
        } else {
          Ud[r] = __builtin_amdgcn_perm (0u, *reinterpret_cast<const u32_any *> (im.p[1] + row * im.s[1] + start), selp);
          Vd[r] = __builtin_amdgcn_perm (0u, *reinterpret_cast<const u32_any *> (im.p[2] + row * im.s[2] + start), selp);
        }
      }
      float hu0[4], hv0[4];
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++) {
        float cu[4], cv[4], hu[4], hv[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { cu[c] = metal::un8 ((Ud[r] >> (8 * c)) & 0xffu); cv[c] = metal::un8 ((Vd[r] >> (8 * c)) & 0xffu); }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          // pixel px + i samples chroma at 0.5 (px + i) - 0.25: weight .75 for even i, .25 for odd i
          const int a = (i + 1) >> 1;                        // index of the first tap in cu[0..3]: i = 0 -> 0, 1 -> 1, 2 -> 1, 3 -> 2
          const float fx = (i & 1) ? 0.25f : 0.75f;
          hu[i] = metal::lerp2 (cu[a], cu[a + 1], fx); hv[i] = metal::lerp2 (cv[a], cv[a + 1], fx);
        }
        // chroma rows (m - 2 + r, m - 1 + r) -> pixel rows: r = 1 -> row 0 (.75); r = 2 -> rows 1 (.25) and 2 (.75); r = 3 -> row 3 (.25)
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int row = 2 * r - 2 - k;                     // r = 1: 0, (-1); r = 2: 2, 1; r = 3: (4), 3
          if (r == 0 || row < 0 || row >= COMP_ROWS) continue;
          const float fy = (row & 1) ? 0.25f : 0.75f;
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float cb = metal::lerp2 (hu0[i], hu[i], fy), cr = metal::lerp2 (hv0[i], hv[i], fy);
            const F4 c = comp_order (metal::yuv_to_rgb (metal::un8 ((Y[row] >> (8 * i)) & 0xffu), cb, cr, im.m709), bgra_out);
            q[row][i] = comp_blend (L, c, q[row][i], flat, bgc);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { hu0[i] = hu[i]; hv0[i] = hv[i]; }
      }
      continue;
    }
    // lane blocks cut by the pad's edge, 4:2:0 pads off the chroma grid: exact-texel fetch pixel by pixel (ONE rolled instance;
    // the register-indexed selects keep q in registers)
#pragma unroll 1
    for (int i = 0; i < 4 * COMP_ROWS; i++) {
      const int c = i & 3, r = i >> 2;
      const bool cxi = c == 0 ? cx[0] : (c == 1 ? cx[1] : (c == 2 ? cx[2] : cx[3]));
      const bool cyi = r == 0 ? cy[0] : (r == 1 ? cy[1] : (r == 2 ? cy[2] : cy[3]));
      if (!(cxi && cyi)) continue;
      const int xi = c == 0 ? xs[0] : (c == 1 ? xs[1] : (c == 2 ? xs[2] : xs[3]));
      const int yi = r == 0 ? ys[0] : (r == 1 ? ys[1] : (r == 2 ? ys[2] : ys[3]));
      uint32_t cur = 0;
#pragma unroll
      for (int j = 0; j < 4 * COMP_ROWS; j++) cur = i == j ? q[j >> 2][j & 3] : cur;
      const uint32_t v = comp_blend (L, comp_order (metal::fetch_1to1 (im, xi - L.xpos, yi - L.ypos, true), bgra_out), cur, flat, bgc);
#pragma unroll
      for (int j = 0; j < 4 * COMP_ROWS; j++) q[j >> 2][j & 3] = i == j ? v : q[j >> 2][j & 3];
    }
  }
  if (!live) return;
  if (p.scratch) {
#pragma unroll
    for (int dy = 0; dy < COMP_ROWS; dy++)
#pragma unroll
      for (int dx = 0; dx < 4; dx++)
        if (4 * bx + dx < p.out.w && COMP_ROWS * by + dy < p.out.h) p.scratch[(size_t) (COMP_ROWS * by + dy) * p.scratch_stride + 4 * bx + dx] = q[dy][dx];
    return;
  }
  if ((o.fmt == VFHIP_FORMAT_BGRA || o.fmt == VFHIP_FORMAT_RGBA) && 4 * bx + 3 < o.w && !(((uintptr_t) o.p[0] | (uintptr_t) o.s[0]) & 15)) {
    typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
#pragma unroll
    for (int dy = 0; dy < COMP_ROWS; dy++) {
      if (COMP_ROWS * by + dy >= o.h) break;
      const v4u v = { q[dy][0], q[dy][1], q[dy][2], q[dy][3] };                              // target order: stored as it is
      __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (o.p[0] + (size_t) (COMP_ROWS * by + dy) * o.s[0]) + bx);
    }
    return;
  }
  // store_block takes logical RGBA (an unaligned or odd-width BGRA output comes this way too)
#pragma unroll
  for (int dy = 0; dy < COMP_ROWS; dy++)
#pragma unroll
    for (int dx = 0; dx < 4; dx++) q[dy][dx] = comp_swap_rb (q[dy][dx], bgra_out);
#pragma unroll
  for (int pr = 0; pr < COMP_PAIRS; pr++) {
    if (2 * (COMP_PAIRS * by + pr) >= o.h) break;
    const uint32_t qa[2][2] = { { q[2 * pr][0], q[2 * pr][1] }, { q[2 * pr + 1][0], q[2 * pr + 1][1] } };
    metal::store_block (o, 2 * bx, COMP_PAIRS * by + pr, qa);
    if (4 * bx + 2 < o.w) {
      const uint32_t qb[2][2] = { { q[2 * pr][2], q[2 * pr][3] }, { q[2 * pr + 1][2], q[2 * pr + 1][3] } };
      metal::store_block (o, 2 * bx + 1, COMP_PAIRS * by + pr, qb);
    }
  }
}


// k_compositor_quads: the lean kernel — the target's start value (a background, or what an RGBA / BGRA output already holds) and up
// to sixteen pads drawn at their own size from RGBA / BGRA frames, into an RGBA / BGRA output whose rows are 16-byte aligned
// (the host checks all of this: comp_quads_ok).  One lane = a 4 x 4 pixel block, as in k_compositor_unscaled, but with neither
// the 4:2:0 sampler nor the scaling sampler in the kernel it needs 61 VGPRs instead of 117-126: 8 waves per SIMD instead of 4,
// and a frame of four 1080p quadrants went from 16.2 to 12.4 us (profiles/r02k_c4_occupancy.txt) — the kernel is a latency-bound
// streaming copy with a blend in the middle, so the bytes in flight per CU are what sets its rate.  Pads of other kinds are drawn
// by the other two kernels as separate launches over their own rectangle (comp_launch), the way the reference draws one quad
// per pad into its render target.
__global__ __launch_bounds__ (256) void k_compositor_quads (const CompParams p)
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const int bx = p.bx0 + blockIdx.x * 64 + threadIdx.x;                                       // 4-pixel column group
  const int by = __builtin_amdgcn_readfirstlane ((int) (p.by0 + blockIdx.y * 4 + threadIdx.y));      // one row of lanes = one wave
  const int x0 = 4 * bx, y0 = COMP_ROWS * by;
  if (y0 >= p.out.h) return;
  const bool live = x0 < p.out.w;                                                             // out.w is a multiple of 4
  const unsigned z = blockIdx.z;
  const metal::OutImg o = metal::out_at (p.out, z * p.out_pitch);
  const bool bgra_out = o.fmt == VFHIP_FORMAT_BGRA;
  const int xl = live ? x0 : p.out.w - 4;                                                     // lanes right of the frame shadow its last block
  bool rowok[COMP_ROWS];
#pragma unroll
  for (int r = 0; r < COMP_ROWS; r++) rowok[r] = y0 + r < p.out.h;                            // wave-uniform
  uint32_t q[COMP_ROWS][4];
  if (p.background == COMP_BG_IN_PLACE) {
#pragma unroll
    for (int r = 0; r < COMP_ROWS; r++) {
      v4u v = { 0u, 0u, 0u, 0u };
      if (rowok[r]) v = *reinterpret_cast<const v4u *> (o.p[0] + (size_t) (y0 + r) * o.s[0] + 4 * xl);
#pragma unroll
      for (int i = 0; i < 4; i++) q[r][i] = v[i];                    // TARGET ORDER: as stored
    }
  } else {
#pragma unroll
    for (int r = 0; r < COMP_ROWS; r++)
#pragma unroll
      for (int i = 0; i < 4; i++) q[r][i] = comp_background (p, o, xl + i, y0 + r);
  }
  const int wx0 = 4 * (p.bx0 + (int) blockIdx.x * 64), wx1 = wx0 + 256, wy1 = y0 + COMP_ROWS;
  if (comp_covered (p, wx0, min (wx1, p.out.w), y0, min (wy1, p.out.h))) return;
  uint32_t hit = 0;                                           // all sixteen rectangle tests up front: independent scalar loads
#pragma unroll
  for (int k = 0; k < COMP_MAX_LAYERS; k++) {
    const CompLayer &L = p.layer[k];
    const bool miss = k >= p.n || wx1 <= L.xpos || wx0 >= L.xpos + L.width || wy1 <= L.ypos || y0 >= L.ypos + L.height;
    hit |= miss ? 0u : 1u << k;
  }
  hit = (uint32_t) __builtin_amdgcn_readfirstlane ((int) hit);
  bool flat = p.background == VFHIP_BG_BLACK || p.background == VFHIP_BG_WHITE || p.background == VFHIP_BG_TRANSPARENT;
  F4 bgc;
  bgc.r = bgc.g = bgc.b = p.background == VFHIP_BG_WHITE ? 1.0f : 0.0f; bgc.a = p.background == VFHIP_BG_TRANSPARENT ? 0.0f : 1.0f;
  for (; hit; flat = false) {
    const int k = __builtin_ctz (hit);
    hit &= hit - 1;
    const CompLayer &L = p.layer[k];
    const uint8_t *base = L.img.p[0] + z * L.pitch;
    const bool rgba_in = (L.img.fmt == VFHIP_FORMAT_RGBA) != bgra_out;        // the pad's texels are in target order
    const int sx = xl - L.xpos, sy = y0 - L.ypos;                  // the lane block's origin in pad coordinates
    if (sx + 3 < 0 || sx >= L.width) continue;                     // (per lane) no column of the block inside the pad
    bool cy[COMP_ROWS];
#pragma unroll
    for (int r = 0; r < COMP_ROWS; r++) cy[r] = rowok[r] && sy + r >= 0 && sy + r < L.height;       // wave-uniform
    v4u t[COMP_ROWS];
    const bool full = sx >= 0 && sx + 3 < L.width;
    if (full) {
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++)                           // every row's load first, then the blends
        if (cy[r]) t[r] = *reinterpret_cast<const v4u __attribute__ ((aligned (4))) *> (base + (size_t) (sy + r) * L.img.s[0] + 4 * sx);
    } else {
      // the lanes a pad's left / right edge cuts: texel by texel, columns clamped into the pad (the uncovered ones are not drawn)
#pragma unroll
      for (int r = 0; r < COMP_ROWS; r++)
        if (cy[r]) {
#pragma unroll
          for (int i = 0; i < 4; i++)
            t[r][i] = *reinterpret_cast<const uint32_t *> (base + (size_t) (sy + r) * L.img.s[0] + 4 * metal::iclamp (sx + i, 0, L.width - 1));
        }
    }
#pragma unroll
    for (int r = 0; r < COMP_ROWS; r++) {
      if (!cy[r]) continue;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t v = comp_blend (L, comp_texel (t[r][i], rgba_in), q[r][i], flat, bgc);
        q[r][i] = (full || (sx + i >= 0 && sx + i < L.width)) ? v : q[r][i];
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int r = 0; r < COMP_ROWS; r++) {
    if (!rowok[r]) break;
    const v4u v = { q[r][0], q[r][1], q[r][2], q[r][3] };
    __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (o.p[0] + (size_t) (y0 + r) * o.s[0]) + bx);
  }
}


// k_compositor_scaled: up to sixteen RGBA / BGRA pads drawn at ANY size (the multiviewer: feeds scaled into the tiles of a mosaic) into an
// RGBA / BGRA output with 16-byte rows.  k_compositor draws a scaled pad through metal::sample_rgba — four plane samples of four
// byte loads each per pixel, the eight pixels of a lane one after the other in a rolled loop (the only way its six-format sampler
// fits the register file): a chain of eight memory latencies per layer, 62 us for a 1080p frame of four down-scaled 1080p feeds.
// Here one lane = 4 x 2 pixels; the tap positions of the four columns and two rows are computed once per layer (metal::lin_taps on
// the same texture coordinate: same operations), a row's sixteen texels (4 pixels x 2 x 2 taps, whole dwords) are loaded together,
// and each pixel is then four byte-wise bilinear interpolations in plane_taps' order -> bit-identical to the general kernel.
// YUV: the pads are NV12 / I420 instead (a multiviewer of decoder feeds): per pixel four luma taps (bytes) and four chroma taps (NV12: (U, V)
// pairs as one 16-bit load; I420: a byte from each plane) at the half-size plane's own tap positions — metal::sample_rgba's plane_sample
// calls with the loads of a row issued together — then metal::yuv_to_rgb.
constexpr int COMP_SROWS = 2;

template <bool YUV>
__global__ __launch_bounds__ (256) void k_compositor_scaled (const CompParams p)
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const int bx = p.bx0 + blockIdx.x * 64 + threadIdx.x;                                       // 4-pixel column group
  const int by = __builtin_amdgcn_readfirstlane ((int) (p.by0 + blockIdx.y * 4 + threadIdx.y));      // one row of lanes = one wave
  const int x0 = 4 * bx, y0 = COMP_SROWS * by;
  if (y0 >= p.out.h) return;
  const int wx0 = 4 * (p.bx0 + (int) blockIdx.x * 64), wx1 = wx0 + 256, wy1 = y0 + COMP_SROWS;
  const bool in_place = p.background == COMP_BG_IN_PLACE;
  if (!in_place && comp_covered (p, wx0, min (wx1, p.out.w), y0, min (wy1, p.out.h))) return;
  const bool live = x0 < p.out.w;                                                             // out.w is a multiple of 4
  const unsigned z = blockIdx.z;
  const metal::OutImg o = metal::out_at (p.out, z * p.out_pitch);
  const bool bgra_out = o.fmt == VFHIP_FORMAT_BGRA;
  const int xl = live ? x0 : p.out.w - 4;                                                     // lanes right of the frame shadow its last block
  bool rowok[COMP_SROWS];
#pragma unroll
  for (int r = 0; r < COMP_SROWS; r++) rowok[r] = y0 + r < p.out.h;                           // wave-uniform
  v4u q[COMP_SROWS];
#pragma unroll
  for (int r = 0; r < COMP_SROWS; r++) {
    q[r] = v4u { 0u, 0u, 0u, 0u };
    if (in_place) { if (rowok[r]) q[r] = *reinterpret_cast<const v4u *> (o.p[0] + (size_t) (y0 + r) * o.s[0] + 4 * xl); }
    else {
#pragma unroll
      for (int i = 0; i < 4; i++) q[r][i] = comp_background (p, o, xl + i, y0 + r);
    }
  }
  uint32_t hit = 0;                                           // all sixteen rectangle tests up front: independent scalar loads
#pragma unroll
  for (int k = 0; k < COMP_MAX_LAYERS; k++) {
    const CompLayer &L = p.layer[k];
    const bool miss = k >= p.n || wx1 <= L.xpos || wx0 >= L.xpos + L.width || wy1 <= L.ypos || y0 >= L.ypos + L.height;
    hit |= miss ? 0u : 1u << k;
  }
  hit = (uint32_t) __builtin_amdgcn_readfirstlane ((int) hit);
  bool flat = p.background == VFHIP_BG_BLACK || p.background == VFHIP_BG_WHITE || p.background == VFHIP_BG_TRANSPARENT;
  F4 bgc;
  bgc.r = bgc.g = bgc.b = p.background == VFHIP_BG_WHITE ? 1.0f : 0.0f; bgc.a = p.background == VFHIP_BG_TRANSPARENT ? 0.0f : 1.0f;
  for (; hit; flat = false) {
    const int k = __builtin_ctz (hit);
    hit &= hit - 1;
    const CompLayer &L = p.layer[k];
    if (xl + 3 < L.xpos || xl >= L.xpos + L.width) continue;       // (per lane) no column of the block inside the quad
    const uint8_t *base = L.img.p[0] + z * L.pitch;
    const int W = L.img.w, H = L.img.h;
    bool cx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cx[i] = xl + i >= L.xpos && xl + i < L.xpos + L.width;
    if (!YUV) {
      const bool swap = (L.img.fmt == VFHIP_FORMAT_BGRA) != bgra_out;  // the pad's bytes 0 and 2 change places on the way into target order
      // the quad's texture coordinate at a pixel centre and its two taps per axis (compositorVertex + the linear sampler: comp_sample)
      uint32_t c0[4], c1[4];                                           // byte offsets of the two tap columns
      float fx[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float tu = (((float) (xl + i) + 0.5f) - (float) L.xpos) / (float) L.width;
        const metal::Taps t = metal::lin_taps (W, tu);
        c0[i] = 4u * (uint32_t) t.i0; c1[i] = 4u * (uint32_t) t.i1; fx[i] = t.f;
      }
#pragma unroll
      for (int r = 0; r < COMP_SROWS; r++) {
        if (!rowok[r] || y0 + r < L.ypos || y0 + r >= L.ypos + L.height) continue;      // wave-uniform
        const float tv = (((float) (y0 + r) + 0.5f) - (float) L.ypos) / (float) L.height;
        const metal::Taps ty = metal::lin_taps (H, tv);
        const uint8_t *r0 = base + (size_t) ty.i0 * L.img.s[0], *r1 = base + (size_t) ty.i1 * L.img.s[0];
        uint32_t t00[4], t10[4], t01[4], t11[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {                               // the row's sixteen texels first (clamped taps: always inside the frame)
          t00[i] = *reinterpret_cast<const uint32_t *> (r0 + c0[i]); t10[i] = *reinterpret_cast<const uint32_t *> (r0 + c1[i]);
          t01[i] = *reinterpret_cast<const uint32_t *> (r1 + c0[i]); t11[i] = *reinterpret_cast<const uint32_t *> (r1 + c1[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {                             // metal::plane_taps per byte: horizontal lerp of each tap row, then vertical
            const float a = metal::lerp2 (metal::un8 ((t00[i] >> (8 * c)) & 0xffu), metal::un8 ((t10[i] >> (8 * c)) & 0xffu), fx[i]);
            const float b = metal::lerp2 (metal::un8 ((t01[i] >> (8 * c)) & 0xffu), metal::un8 ((t11[i] >> (8 * c)) & 0xffu), fx[i]);
            v[c] = metal::lerp2 (a, b, ty.f);
          }
          F4 sc;
          sc.r = swap ? v[2] : v[0]; sc.g = v[1]; sc.b = swap ? v[0] : v[2]; sc.a = v[3];      // target order
          const uint32_t nv = comp_blend (L, sc, q[r][i], flat, bgc);
          q[r][i] = cx[i] ? nv : q[r][i];
        }
      }
    } else {
      typedef uint16_t __attribute__ ((aligned (1))) u16_any;
      const bool nv12 = L.img.fmt == VFHIP_FORMAT_NV12;
      const uint8_t *cb_base = L.img.p[1] + z * L.pitch, *cr_base = nv12 ? cb_base : L.img.p[2] + z * L.pitch;
      const int cw = (W + 1) / 2, chh = (H + 1) / 2;
      const uint32_t cstep = nv12 ? 2u : 1u;
      uint32_t y0c[4], y1c[4], k0c[4], k1c[4];                        // byte offsets of the luma / chroma tap columns
      float fy_[4], fk[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float tu = (((float) (xl + i) + 0.5f) - (float) L.xpos) / (float) L.width;
        const metal::Taps t = metal::lin_taps (W, tu), k = metal::lin_taps (cw, tu);
        y0c[i] = (uint32_t) t.i0; y1c[i] = (uint32_t) t.i1; fy_[i] = t.f;
        k0c[i] = cstep * (uint32_t) k.i0; k1c[i] = cstep * (uint32_t) k.i1; fk[i] = k.f;
      }
#pragma unroll
      for (int r = 0; r < COMP_SROWS; r++) {
        if (!rowok[r] || y0 + r < L.ypos || y0 + r >= L.ypos + L.height) continue;      // wave-uniform
        const float tv = (((float) (y0 + r) + 0.5f) - (float) L.ypos) / (float) L.height;
        const metal::Taps ty = metal::lin_taps (H, tv), tk = metal::lin_taps (chh, tv);
        const uint8_t *l0 = base + (size_t) ty.i0 * L.img.s[0], *l1 = base + (size_t) ty.i1 * L.img.s[0];
        const uint8_t *u0 = cb_base + (size_t) tk.i0 * L.img.s[1], *u1 = cb_base + (size_t) tk.i1 * L.img.s[1];
        const uint8_t *v0 = cr_base + (size_t) tk.i0 * L.img.s[nv12 ? 1 : 2], *v1 = cr_base + (size_t) tk.i1 * L.img.s[nv12 ? 1 : 2];
        uint32_t ya[4], yb[4], yc[4], yd[4], ca[4], cb_[4], cc[4], cd[4];      // c*: (U | V << 8) of the four chroma taps
#pragma unroll
        for (int i = 0; i < 4; i++) {                               // the row's loads first (clamped taps: always inside the planes)
          ya[i] = l0[y0c[i]]; yb[i] = l0[y1c[i]]; yc[i] = l1[y0c[i]]; yd[i] = l1[y1c[i]];
          if (nv12) {
            ca[i] = *reinterpret_cast<const u16_any *> (u0 + k0c[i]); cb_[i] = *reinterpret_cast<const u16_any *> (u0 + k1c[i]);
            cc[i] = *reinterpret_cast<const u16_any *> (u1 + k0c[i]); cd[i] = *reinterpret_cast<const u16_any *> (u1 + k1c[i]);
          } else {
            ca[i] = (uint32_t) u0[k0c[i]] | ((uint32_t) v0[k0c[i]] << 8); cb_[i] = (uint32_t) u0[k1c[i]] | ((uint32_t) v0[k1c[i]] << 8);
            cc[i] = (uint32_t) u1[k0c[i]] | ((uint32_t) v1[k0c[i]] << 8); cd[i] = (uint32_t) u1[k1c[i]] | ((uint32_t) v1[k1c[i]] << 8);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          using metal::lerp2; using metal::un8;
          const float yy = lerp2 (lerp2 (un8 (ya[i]), un8 (yb[i]), fy_[i]), lerp2 (un8 (yc[i]), un8 (yd[i]), fy_[i]), ty.f);
          const float cb = lerp2 (lerp2 (un8 (ca[i] & 0xffu), un8 (cb_[i] & 0xffu), fk[i]), lerp2 (un8 (cc[i] & 0xffu), un8 (cd[i] & 0xffu), fk[i]), tk.f);
          const float cr = lerp2 (lerp2 (un8 (ca[i] >> 8), un8 (cb_[i] >> 8), fk[i]), lerp2 (un8 (cc[i] >> 8), un8 (cd[i] >> 8), fk[i]), tk.f);
          const uint32_t nv = comp_blend (L, comp_order (metal::yuv_to_rgb (yy, cb, cr, L.img.m709), bgra_out), q[r][i], flat, bgc);
          q[r][i] = cx[i] ? nv : q[r][i];
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int r = 0; r < COMP_SROWS; r++) {
    if (!rowok[r]) break;
    __builtin_nontemporal_store (q[r], reinterpret_cast<v4u *> (o.p[0] + (size_t) (y0 + r) * o.s[0]) + bx);
  }
}

// k_compositor_420: ONE pad drawn at its own size from an NV12 / I420 frame on the chroma grid (even xpos / ypos: comp_pad_420) into an
// RGBA / BGRA output with 16-byte rows — a camera or decoder feed in a mosaic, the inset of BASELINE configs[3].
// k_compositor_unscaled draws such a pad from a 4 x 4 block per lane with all sixteen running colours in registers (117 VGPRs, 4 waves
// per SIMD, 17.2 us for a frame of four 1080p NV12 quadrants where the traffic needs 7.4); with one pad there is nothing to carry
// between pads, so this kernel WALKS a strip of COMP420_ROWS rows, two rows (one chroma step) per trip: a row's four target pixels are
// loaded, blended and stored, and the only state is the two horizontally interpolated chroma rows the next trip starts from.
// Same operations per value as metal::fetch_1to1 / plane_taps (horizontal lerp of each chroma row, then the vertical lerp, .25 / .75
// phases) -> bit-identical to the other kernels.  Rows and lanes outside the pad are not touched when drawing in place; an opaque pad
// (alpha 1, not ADD: out = s exactly, k = 1 - 1 = 0) does not read the target under it at all.
// Lanes that a pad edge cuts (xpos or width not a multiple of 4) take the exact-texel sampler pixel by pixel.
constexpr int COMP420_ROWS = 8;

struct C420Row { float u[4], v[4]; };
struct C420Raw { uint32_t a, b; };      // a chroma row's four columns as loaded: NV12 two dwords of (U, V) pairs, I420 the U and the V dword
// chroma row `row` (clamped by the caller), columns start .. start + 3 (start: j - 1 clamped into the plane: see k_compositor_unscaled)
__device__ __forceinline__ C420Raw comp420_load (const metal::Img &im, size_t row, int start)
{
  typedef uint32_t __attribute__ ((aligned (1))) u32_any;
  typedef uint2 __attribute__ ((aligned (1))) u64_any;
  C420Raw r;
  if (im.fmt == VFHIP_FORMAT_NV12) {
    const uint2 c = *reinterpret_cast<const u64_any *> (im.p[1] + row * im.s[1] + 2 * start);
    r.a = c.x; r.b = c.y;
  } else {
    r.a = *reinterpret_cast<const u32_any *> (im.p[1] + row * im.s[1] + start);
    r.b = *reinterpret_cast<const u32_any *> (im.p[2] + row * im.s[2] + start);
  }
  return r;
}
// ... -> columns j - 1 .. j + 2 edge-clamped by the byte selectors -> the four pixels' horizontally interpolated (U, V)
__device__ __forceinline__ C420Row comp420_hrow (const C420Raw &c, bool nv12, uint32_t selu, uint32_t selp)
{
  const uint32_t Ud = nv12 ? __builtin_amdgcn_perm (c.b, c.a, selu) : __builtin_amdgcn_perm (0u, c.a, selp);
  const uint32_t Vd = nv12 ? __builtin_amdgcn_perm (c.b, c.a, selu + 0x01010101u) : __builtin_amdgcn_perm (0u, c.b, selp);
  float cu[4], cv[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { cu[k] = metal::un8 ((Ud >> (8 * k)) & 0xffu); cv[k] = metal::un8 ((Vd >> (8 * k)) & 0xffu); }
  C420Row h;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int t = (i + 1) >> 1;                            // pixel px + i: taps (t, t + 1) of the four columns, weight .75 (even i) / .25 (odd i)
    const float fx = (i & 1) ? 0.25f : 0.75f;
    h.u[i] = metal::lerp2 (cu[t], cu[t + 1], fx); h.v[i] = metal::lerp2 (cv[t], cv[t + 1], fx);
  }
  return h;
}

// clamp01 (fmaf (a, b, c)) in one instruction (the clamp output modifier of v_fma_f32; see deinterlace.hip)
__device__ __forceinline__ float comp_fma_sat (float a, float b, float c)
{
  float d;
  asm ("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// TARGET: the pad's pixels are blended with what the target holds (read back pixel by pixel); false: with one wave-uniform colour — the
// flat background of a frame's first run, or nothing at all under an opaque pad (k = 0) — so the fast path loads no target, converts
// none, and the alpha it writes is a constant.  Everything the per-pixel code would branch on (matrix, byte order, operator) is
// turned into wave-uniform coefficients before the loop, and the loads of a row pair are issued one trip ahead (row and chroma
// indices clamped, so the prefetch needs no branch): the loop body is straight-line code with its inputs already on their way.
template <bool TARGET>
__global__ __launch_bounds__ (256) void k_compositor_420 (const CompParams p)
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  typedef uint32_t __attribute__ ((aligned (1))) u32_any;
  const int bx = p.bx0 + blockIdx.x * 64 + threadIdx.x;
  const int by = __builtin_amdgcn_readfirstlane ((int) (p.by0 + blockIdx.y * 4 + threadIdx.y));      // one row of lanes = one wave
  const int x0 = 4 * bx, y0 = COMP420_ROWS * by;
  if (y0 >= p.out.h) return;
  const int y1 = min (y0 + COMP420_ROWS, p.out.h);
  const bool live = x0 < p.out.w;                              // out.w is a multiple of 4
  const int xl = live ? x0 : p.out.w - 4;
  const unsigned z = blockIdx.z;
  const metal::OutImg o = metal::out_at (p.out, z * p.out_pitch);
  const bool bgra_out = o.fmt == VFHIP_FORMAT_BGRA;
  const bool in_place = p.background == COMP_BG_IN_PLACE;
  const int wx0 = 4 * (p.bx0 + (int) blockIdx.x * 64);
  if (!in_place && comp_covered (p, wx0, min (wx0 + 256, p.out.w), y0, y1)) return;
  const CompLayer &L = p.layer[0];
  const metal::Img im = metal::img_at (L.img, z * L.pitch);
  const bool nv12 = im.fmt == VFHIP_FORMAT_NV12;
  // the strip's rows under the pad: [ya, yb) (wave-uniform); ya - ypos is even because ypos and the strip height are
  const int ya = max (y0, L.ypos), yb = min (y1, L.ypos + L.height);
  uint8_t *orow = o.p[0] + 4 * (size_t) xl;
  if (!in_place && live) {
    // a frame's first run: the strip's rows above and below the pad are background
    for (int y = y0; y < y1; y++) {
      if (y >= ya && y < yb) continue;
      v4u b;
#pragma unroll
      for (int i = 0; i < 4; i++) b[i] = comp_background (p, o, xl + i, y);
      __builtin_nontemporal_store (b, reinterpret_cast<v4u *> (orow + (size_t) y * o.s[0]));
    }
  }
  if (ya >= yb) return;
  const bool flat = !in_place && (p.background == VFHIP_BG_BLACK || p.background == VFHIP_BG_WHITE || p.background == VFHIP_BG_TRANSPARENT);
  F4 bgc;
  bgc.r = bgc.g = bgc.b = p.background == VFHIP_BG_WHITE ? 1.0f : 0.0f; bgc.a = p.background == VFHIP_BG_TRANSPARENT ? 0.0f : 1.0f;
  if (in_place) bgc.r = bgc.g = bgc.b = bgc.a = 0.0f;           // (TARGET false in place: an opaque pad, the target is multiplied by k = 0)
  const int sx = xl - L.xpos;                                   // even (comp_pad_420)
  const bool anyx = sx + 3 >= 0 && sx < L.width, full = sx >= 0 && sx + 3 < L.width;
  if (!full) {
    // lanes beside the pad (a first run gives them the background) and lanes its left / right edge cuts: the target under them, then
    // the exact-texel sampler for the covered pixels, one at a time (one rolled instance)
    if (in_place && !anyx) return;
#pragma unroll 1
    for (int y = ya; y < yb; y++) {
      v4u q;
      if (in_place) q = *reinterpret_cast<const v4u *> (orow + (size_t) y * o.s[0]);
      else {
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = comp_background (p, o, xl + i, y);
      }
      if (anyx) {
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
          if (sx + i < 0 || sx + i >= L.width) continue;
          const uint32_t cur = i == 0 ? q[0] : (i == 1 ? q[1] : (i == 2 ? q[2] : q[3]));
          const uint32_t v = comp_blend (L, comp_order (metal::fetch_1to1 (im, sx + i, y - L.ypos, true), bgra_out), cur, flat, bgc);
          q[0] = i == 0 ? v : q[0]; q[1] = i == 1 ? v : q[1]; q[2] = i == 2 ? v : q[2]; q[3] = i == 3 ? v : q[3];
        }
      }
      if (live) __builtin_nontemporal_store (q, reinterpret_cast<v4u *> (orow + (size_t) y * o.s[0]));
    }
    return;
  }
  // ---- lanes inside the pad ----
  // the blend of a pixel without alpha of its own (c.a = 1): s.a = 1 * alpha = alpha, s.rgb = c.rgb * alpha, out = d * k + s with
  // k = 0 (source), 1 (add), 1 - alpha (over) — comp_blend_f's operations with the wave-uniform ones done once
  const float alpha = L.alpha;
  const float kb = L.blend == VFHIP_BLEND_SOURCE ? 0.0f : (L.blend == VFHIP_BLEND_ADD ? 1.0f : 1.0f - alpha);
  const bool m709 = im.m709 != 0;
  const float c_rv = m709 ? 1.792741f : 1.596027f, c_gu = m709 ? -0.213249f : -0.391762f, c_gv = m709 ? -0.532909f : -0.812968f,
              c_bu = m709 ? 2.112402f : 2.017232f;           // metal::yuv_to_rgb's two matrices
  const uint32_t qa_flat = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (bgc.a, kb, alpha) * 255.0f, 3u, 0u);      // TARGET false: the alpha byte, in place
  const int cw = (im.w + 1) >> 1, chh = (im.h + 1) >> 1;
  const int j = sx >> 1, start = metal::iclamp (j - 1, 0, cw - 4);
  const uint32_t selu = j < 1 ? 0x04020000u : (j + 2 >= cw ? 0x06060402u : 0x06040200u);
  const uint32_t selp = j < 1 ? 0x02010000u : (j + 2 >= cw ? 0x03030201u : 0x03020100u);
  const int sy0 = ya - L.ypos, m0 = sy0 >> 1;                   // first pad row of the strip (even), its chroma row
  const uint8_t *yrow = im.p[0] + sx;
  // prologue: chroma rows m0 - 1 and m0, and the first pair's loads
  const C420Raw ra = comp420_load (im, (size_t) max (m0 - 1, 0), start), rb = comp420_load (im, (size_t) min (m0, chh - 1), start);
  uint32_t Yn[2];
  C420Raw rn;
  v4u qn[2] = { { 0u, 0u, 0u, 0u }, { 0u, 0u, 0u, 0u } };
  auto prefetch = [&] (int sy, int ye) {
#pragma unroll
    for (int r = 0; r < 2; r++) Yn[r] = *reinterpret_cast<const u32_any *> (yrow + (size_t) min (sy + r, L.height - 1) * im.s[0]);
    rn = comp420_load (im, (size_t) min ((sy >> 1) + 1, chh - 1), start);
    if (TARGET && in_place) {
#pragma unroll
      for (int r = 0; r < 2; r++) qn[r] = *reinterpret_cast<const v4u *> (orow + (size_t) min (ye + r, p.out.h - 1) * o.s[0]);
    }
  };
  prefetch (sy0, ya);
  C420Row hA = comp420_hrow (ra, nv12, selu, selp), hB = comp420_hrow (rb, nv12, selu, selp);
#pragma unroll 1
  for (int ye = ya; ye < yb; ye += 2) {
    const int sy = ye - L.ypos;                                 // even
    const uint32_t Y[2] = { Yn[0], Yn[1] };
    const C420Raw rc = rn;
    v4u q[2] = { qn[0], qn[1] };
    prefetch (min (sy + 2, L.height - 1) & ~1, min (ye + 2, p.out.h - 1));       // the next pair's (the last trip re-reads its own: no branch)
    if (TARGET && !in_place) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) q[r][i] = comp_background (p, o, xl + i, min (ye + r, p.out.h - 1));
    }
    const C420Row hC = comp420_hrow (rc, nv12, selu, selp);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      // row 2m leans on chroma rows (m - 1, m) with .75, row 2m + 1 on (m, m + 1) with .25
      const C420Row &c0 = r ? hB : hA, &c1 = r ? hC : hB;
      const float fy = r ? 0.25f : 0.75f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float cb = metal::lerp2 (c0.u[i], c1.u[i], fy), cr = metal::lerp2 (c0.v[i], c1.v[i], fy);
        // metal::yuv_to_rgb, the matrix in registers and each clamp folded into the fma that feeds it
        const float yy = metal::un8 ((Y[r] >> (8 * i)) & 0xffu) - 16.0f / 255.0f, u = cb - 128.0f / 255.0f, v = cr - 128.0f / 255.0f;
        const float ly = 1.164383f * yy;
        const float cr_ = comp_fma_sat (c_rv, v, ly), cg_ = comp_fma_sat (c_gv, v, fmaf (c_gu, u, ly)), cb_ = comp_fma_sat (c_bu, u, ly);
        const float s0 = (bgra_out ? cb_ : cr_) * alpha, s1 = cg_ * alpha, s2 = (bgra_out ? cr_ : cb_) * alpha;      // target order, premultiplied
        uint32_t v8;
        if (TARGET) {
          const F4 d = metal::unpack_rgba8 (q[r][i]);
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (d.r, kb, s0) * 255.0f, 0u, 0u);
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (d.g, kb, s1) * 255.0f, 1u, v8);
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (d.b, kb, s2) * 255.0f, 2u, v8);
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (d.a, kb, alpha) * 255.0f, 3u, v8);
        } else {
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (bgc.r, kb, s0) * 255.0f, 0u, qa_flat);
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (bgc.g, kb, s1) * 255.0f, 1u, v8);
          v8 = __builtin_amdgcn_cvt_pk_u8_f32 (fmaf (bgc.b, kb, s2) * 255.0f, 2u, v8);
        }
        q[r][i] = v8;
      }
    }
    hA = hB; hB = hC;
    if (live) {
      __builtin_nontemporal_store (q[0], reinterpret_cast<v4u *> (orow + (size_t) ye * o.s[0]));
      if (ye + 1 < yb) __builtin_nontemporal_store (q[1], reinterpret_cast<v4u *> (orow + (size_t) (ye + 1) * o.s[0]));
    }
  }
}

}  // namespace vfhip

struct VfHipCompositor {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo out {};
  uint32_t *scratch[2] = { nullptr, nullptr };   // only for > COMP_MAX_LAYERS pads
  Flights fl;                                    // _submit / _wait: flight k owns staging slots k * COMP_FLIGHT_SLOTS ...
};

static bool comp_pad_visible (const VfHipPadInput &in) { return in.width > 0 && in.height > 0; }
static bool comp_pad_unscaled (const VfHipPadInput &in) { return in.width == in.frame.info.width && in.height == in.frame.info.height; }
// a pad k_compositor_quads can draw
static bool comp_pad_lean (const VfHipPadInput &in)
{
  return comp_pad_unscaled (in) && (in.frame.info.format == VFHIP_FORMAT_RGBA || in.frame.info.format == VFHIP_FORMAT_BGRA) &&
      !(((uintptr_t) in.frame.data[0] | (uintptr_t) in.frame.stride[0]) & 3);
}
// a pad k_compositor_420 can draw: 4:2:0 at its own size, on the chroma grid of the output's lane blocks
static bool comp_pad_420 (const VfHipPadInput &in)
{
  return comp_pad_unscaled (in) && (in.frame.info.format == VFHIP_FORMAT_NV12 || in.frame.info.format == VFHIP_FORMAT_I420) &&
      !((in.xpos | in.ypos) & 1) && in.frame.info.width >= 7;
}
// what is under this pad does not show: it replaces the target (alpha 1 and a format without alpha under OVER; SOURCE always)
static bool comp_pad_overwrites (const VfHipPadInput &in)
{
  const int f = in.frame.info.format;
  if (in.blend_mode == VFHIP_BLEND_SOURCE) return true;
  return in.blend_mode == VFHIP_BLEND_OVER && (float) in.alpha == 1.0f && f != VFHIP_FORMAT_RGBA && f != VFHIP_FORMAT_BGRA;
}
// a pad k_compositor_scaled can draw: RGBA / BGRA, scaled (a pad at its own size is sampled at exact texels: comp_sample)
static bool comp_pad_rgba (const VfHipPadInput &in)
{
  return !comp_pad_unscaled (in) && (in.frame.info.format == VFHIP_FORMAT_RGBA || in.frame.info.format == VFHIP_FORMAT_BGRA) &&
      !(((uintptr_t) in.frame.data[0] | (uintptr_t) in.frame.stride[0]) & 3);
}
// a pad k_compositor_scaled<true> can draw: NV12 / I420, scaled
static bool comp_pad_yuv_scaled (const VfHipPadInput &in)
{
  return !comp_pad_unscaled (in) && (in.frame.info.format == VFHIP_FORMAT_NV12 || in.frame.info.format == VFHIP_FORMAT_I420);
}
enum { COMP_KIND_LEAN, COMP_KIND_420, COMP_KIND_SCALED, COMP_KIND_SCALED_YUV, COMP_KIND_HEAVY };

static void comp_fill_layer (CompLayer &L, const VfHipPadInput &in, size_t pitch)
{
  L.img = metal::make_img (&in.frame);
  L.xpos = in.xpos; L.ypos = in.ypos; L.width = in.width; L.height = in.height;
  L.alpha = (float) in.alpha; L.blend = in.blend_mode;
  L.pitch = pitch;
}

// RGBA / BGRA output with 16-byte rows: the output itself is the render target, and the pads are drawn in RUNS — consecutive pads
// of one kind (`lean`: what k_compositor_quads draws; the rest: the samplers of k_compositor_unscaled / k_compositor), at most
// COMP_MAX_LAYERS each.  The first run's launch covers the frame and starts from the background; every later run is a launch over
// the bounding rectangle of its pads that starts from what the output holds (COMP_BG_IN_PLACE) — one draw per run where the
// reference has one draw per pad (metalcomprenderer.m:356-542), the same 8-bit target between them.  A lane reads and writes only
// its own pixels, so drawing in place needs no second buffer.
static int comp_launch_runs (VfHipCompositor *h, const VfHipPadInput *pads, int count, int background, VfHipFrame *out, hipStream_t s,
    int n_frames, const size_t *pad_pitch, size_t out_pitch)
{
  const int w = h->out.width, hh = h->out.height;
  const bool force_general = getenv ("VFHIP_COMP_GENERAL") != nullptr;       // test knobs: one kernel for every run, ...
  const bool no_lean = force_general || getenv ("VFHIP_COMP_NO_QUADS") != nullptr;       // ... no k_compositor_quads, ...
  const bool no_420 = force_general || getenv ("VFHIP_COMP_NO_420") != nullptr;          // ... no k_compositor_420, ...
  const bool no_scaled = force_general || getenv ("VFHIP_COMP_NO_SCALED") != nullptr;    // ... no k_compositor_scaled, ...
  const bool no_cover = getenv ("VFHIP_COMP_NO_COVER") != nullptr;                       // ... draw what a later opaque pad hides
  auto kind_of = [&] (const VfHipPadInput &in) {
    if (!no_lean && comp_pad_lean (in)) return (int) COMP_KIND_LEAN;
    if (!no_420 && comp_pad_420 (in)) return (int) COMP_KIND_420;
    if (!no_scaled && comp_pad_rgba (in)) return (int) COMP_KIND_SCALED;
    if (!no_scaled && comp_pad_yuv_scaled (in)) return (int) COMP_KIND_SCALED_YUV;
    return (int) COMP_KIND_HEAVY;
  };
  int k = 0;
  bool first = true;
  while (k < count || first) {
    while (k < count && !comp_pad_visible (pads[k])) k++;
    CompParams p {};
    int kind = no_lean ? COMP_KIND_HEAVY : COMP_KIND_LEAN;
    bool unscaled = true;
    int rx0 = w, ry0 = hh, rx1 = 0, ry1 = 0;
    if (k < count) {
      kind = kind_of (pads[k]);
      for (; k < count && p.n < (kind == COMP_KIND_420 ? 1 : COMP_MAX_LAYERS); k++) {
        const VfHipPadInput &in = pads[k];
        if (!comp_pad_visible (in)) continue;
        if (kind_of (in) != kind) break;
        unscaled = unscaled && comp_pad_unscaled (in);
        comp_fill_layer (p.layer[p.n++], in, pad_pitch ? pad_pitch[k] : 0);
        rx0 = std::min (rx0, in.xpos); ry0 = std::min (ry0, in.ypos);
        rx1 = std::max (rx1, (int) std::min ((long long) w, (long long) in.xpos + in.width));
        ry1 = std::max (ry1, (int) std::min ((long long) hh, (long long) in.ypos + in.height));
      }
    }
    if (kind == COMP_KIND_420 && p.n == 0) kind = COMP_KIND_LEAN;            // (no pad left: a background-only launch)
    if (first) {
      rx0 = 0; ry0 = 0; rx1 = w; ry1 = hh;
      // what later opaque pads overwrite completely need not be drawn by this launch (largest rectangles first)
      if (!no_cover && kind != COMP_KIND_HEAVY) {      // (the three run kernels test the rectangles; the samplers' kernels draw everything)
        struct R { int x0, y0, x1, y1; long long area; } best[COMP_MAX_COVER];
        int nb = 0;
        for (int j = k; j < count; j++) {
          const VfHipPadInput &in = pads[j];
          if (!comp_pad_visible (in) || !comp_pad_overwrites (in)) continue;
          R r { std::max (in.xpos, 0), std::max (in.ypos, 0), (int) std::min ((long long) w, (long long) in.xpos + in.width),
                (int) std::min ((long long) hh, (long long) in.ypos + in.height), 0 };
          if (r.x0 >= r.x1 || r.y0 >= r.y1) continue;
          r.area = (long long) (r.x1 - r.x0) * (r.y1 - r.y0);
          int at = nb < COMP_MAX_COVER ? nb++ : -1;
          if (at < 0) { for (int q = 0; q < nb; q++) if (best[q].area < r.area && (at < 0 || best[q].area < best[at].area)) at = q; }
          if (at >= 0) best[at] = r;
        }
        p.n_cover = nb;
        for (int q = 0; q < nb; q++) { p.cover[q][0] = best[q].x0; p.cover[q][1] = best[q].y0; p.cover[q][2] = best[q].x1; p.cover[q][3] = best[q].y1; }
      }
    }
    rx0 = std::max (rx0, 0); ry0 = std::max (ry0, 0);
    const bool draw = first || (p.n > 0 && rx0 < rx1 && ry0 < ry1);
    p.background = first ? background : COMP_BG_IN_PLACE;
    first = false;
    if (!draw) continue;
    unscaled = unscaled && !force_general;
    const int rows = kind == COMP_KIND_420 ? COMP420_ROWS : ((kind == COMP_KIND_SCALED || kind == COMP_KIND_SCALED_YUV) ? COMP_SROWS : ((kind == COMP_KIND_LEAN || unscaled) ? COMP_ROWS : 2));
    // whole 64-lane groups from a 256-pixel boundary keep the 16-byte lanes of a wave on one 1 KiB-aligned run of a row
    p.bx0 = (rx0 / 256) * 64; p.by0 = ry0 / rows;
    const int bx1 = (rx1 + 3) / 4, by1 = (ry1 + rows - 1) / rows;
    dim3 grid ((unsigned) ((bx1 - p.bx0 + 63) / 64), (unsigned) ((by1 - p.by0 + 3) / 4), (unsigned) n_frames);
    p.out = metal::make_out (out); p.out_pitch = out_pitch;
    if (kind == COMP_KIND_LEAN) hipLaunchKernelGGL (k_compositor_quads, grid, dim3 (64, 4), 0, s, p);
    else if (kind == COMP_KIND_420) {
      // nothing to read back under the pad: an opaque pad drawn in place (out = s), or a frame's first run over a uniform background
      const bool flat_bg = p.background == VFHIP_BG_BLACK || p.background == VFHIP_BG_WHITE || p.background == VFHIP_BG_TRANSPARENT;
      const bool opaque = p.layer[0].alpha == 1.0f && p.layer[0].blend != VFHIP_BLEND_ADD;
      if (p.background == COMP_BG_IN_PLACE ? opaque : flat_bg) hipLaunchKernelGGL (k_compositor_420<false>, grid, dim3 (64, 4), 0, s, p);
      else hipLaunchKernelGGL (k_compositor_420<true>, grid, dim3 (64, 4), 0, s, p);
    }
    else if (kind == COMP_KIND_SCALED) hipLaunchKernelGGL (k_compositor_scaled<false>, grid, dim3 (64, 4), 0, s, p);
    else if (kind == COMP_KIND_SCALED_YUV) hipLaunchKernelGGL (k_compositor_scaled<true>, grid, dim3 (64, 4), 0, s, p);
    else if (unscaled) hipLaunchKernelGGL (k_compositor_unscaled, grid, dim3 (64, 4), 0, s, p);
    else hipLaunchKernelGGL (k_compositor, grid, dim3 (64, 4), 0, s, p);
    VFHIP_CHECK_HIP (hipGetLastError ());
  }
  return VFHIP_OK;
}

static int comp_launch (VfHipCompositor *h, const VfHipPadInput *pads, int count, int background, VfHipFrame *out, hipStream_t s,
    int n_frames = 1, const size_t *pad_pitch = nullptr, size_t out_pitch = 0)
{
  const int w = h->out.width, hh = h->out.height;
  const int ofmt = out->info.format;
  if ((ofmt == VFHIP_FORMAT_RGBA || ofmt == VFHIP_FORMAT_BGRA) && !(w & 3) &&
      !(((uintptr_t) out->data[0] | (uintptr_t) out->stride[0] | (uintptr_t) out_pitch) & 15) && !getenv ("VFHIP_COMP_ONE_PASS"))
    return comp_launch_runs (h, pads, count, background, out, s, n_frames, pad_pitch, out_pitch);
  // planar / odd-sized outputs: one launch walks all the pads (up to COMP_MAX_LAYERS; more chain through an RGBA8 scratch target)
  // every pad at its own size -> k_compositor_unscaled (4 x 4 pixel blocks per lane); otherwise the general kernel (4 x 2)
  bool unscaled = getenv ("VFHIP_COMP_GENERAL") == nullptr;       // test knob: force the general kernel
  for (int k = 0; k < count && unscaled; k++)
    if (comp_pad_visible (pads[k]) && !comp_pad_unscaled (pads[k])) unscaled = false;
  const int rows = unscaled ? COMP_ROWS : 2;
  const int bw = (w + 3) / 4, bh = (hh + rows - 1) / rows;
  dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
  const int passes = count <= COMP_MAX_LAYERS ? 1 : (count + COMP_MAX_LAYERS - 1) / COMP_MAX_LAYERS;
  if (passes > 1)
    for (int k = 0; k < 2; k++)
      if (!h->scratch[k]) VFHIP_CHECK_HIP (dev_malloc (&h->scratch[k], (size_t) w * hh * 4));
  for (int pass = 0; pass < passes; pass++) {
    CompParams p {};
    const int first = pass * COMP_MAX_LAYERS, n = count - first < COMP_MAX_LAYERS ? count - first : COMP_MAX_LAYERS;
    p.n = 0;
    for (int k = 0; k < n; k++)
      if (comp_pad_visible (pads[first + k])) comp_fill_layer (p.layer[p.n++], pads[first + k], pad_pitch ? pad_pitch[first + k] : 0);
    p.background = pass == 0 ? background : -1;
    p.prev = pass == 0 ? nullptr : h->scratch[(pass - 1) & 1]; p.prev_stride = w;
    p.out = metal::make_out (out); p.out_pitch = out_pitch;
    p.scratch = pass == passes - 1 ? nullptr : h->scratch[pass & 1]; p.scratch_stride = w;
    if (unscaled) hipLaunchKernelGGL (k_compositor_unscaled, grid, dim3 (64, 4), 0, s, p);
    else hipLaunchKernelGGL (k_compositor, grid, dim3 (64, 4), 0, s, p);
    VFHIP_CHECK_HIP (hipGetLastError ());
  }
  return VFHIP_OK;
}

static int comp_check (VfHipCompositor *h, const VfHipPadInput *pads, int count, int background, const VfHipFrame *out)
{
  if (!h || (count > 0 && !pads)) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "compositor: composite before configure");
  if (count < 0 || count > 4096) return set_error (VFHIP_ERR_INVALID, "bad pad count %d", count);
  if (background < VFHIP_BG_CHECKER || background > VFHIP_BG_TRANSPARENT) return set_error (VFHIP_ERR_INVALID, "bad background %d", background);
  for (int k = 0; k < count; k++) {
    int rc = check_frame (&pads[k].frame, nullptr, "pad");
    if (rc) return rc;
    if (pads[k].frame.info.format > VFHIP_FORMAT_I420) return set_error (VFHIP_ERR_UNSUPPORTED, "pad %d: format not supported", k);
    if (pads[k].blend_mode < VFHIP_BLEND_SOURCE || pads[k].blend_mode > VFHIP_BLEND_ADD) return set_error (VFHIP_ERR_INVALID, "pad %d: bad blend mode", k);
  }
  return check_frame (out, &h->out, "output");
}

extern "C" {

VfHipCompositor *vfhip_compositor_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipCompositor *h = new (std::nothrow) VfHipCompositor ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_compositor_configure (VfHipCompositor *h, const VfHipVideoInfo *out)
{
  if (!h || !out) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted composite(s) still in flight: wait for them first", h->fl.count);
  if (out->width <= 0 || out->height <= 0 || out->width > 32768 || out->height > 32768)
    return set_error (VFHIP_ERR_INVALID, "bad output size %dx%d", out->width, out->height);
  if (out->format < VFHIP_FORMAT_BGRA || out->format > VFHIP_FORMAT_I420)
    return set_error (VFHIP_ERR_UNSUPPORTED, "compositor: output format not supported");
  (void) hipSetDevice (h->dev->ordinal);
  for (int k = 0; k < 2; k++) { if (h->scratch[k]) (void) hipFree (h->scratch[k]); h->scratch[k] = nullptr; }
  h->out = *out; h->configured = true;
  return VFHIP_OK;
}

int vfhip_compositor_composite_device (VfHipCompositor *h, const VfHipPadInput *pads, int count, int background,
    VfHipFrame *out, void *stream)
{
  int rc = comp_check (h, pads, count, background, out);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return comp_launch (h, pads, count, background, out, stream ? (hipStream_t) stream : h->st.s_compute);
}

int vfhip_compositor_composite_device_batch (VfHipCompositor *h, const VfHipPadInput *pads, const size_t *pad_frame_pitch, int count,
    int background, VfHipFrame *out0, size_t out_frame_pitch, int n_frames, void *stream)
{
  int rc = comp_check (h, pads, count, background, out0);
  if (rc) return rc;
  if (n_frames < 1 || n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "n_frames %d outside 1..65535", n_frames);
  if (n_frames > 1 && (count > COMP_MAX_LAYERS || !pad_frame_pitch))
    return set_error (VFHIP_ERR_UNSUPPORTED, "batched compositing takes at most %d pads and needs the per-pad frame pitches", COMP_MAX_LAYERS);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return comp_launch (h, pads, count, background, out0, stream ? (hipStream_t) stream : h->st.s_compute, n_frames, pad_frame_pitch, out_frame_pitch);
}

int vfhip_compositor_composite (VfHipCompositor *h, const VfHipPadInput *pads, int count, int background, VfHipFrame *out)
{
  int rc = comp_check (h, pads, count, background, out);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_compositor_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  std::vector<VfHipPadInput> dpads (pads, pads + count);
  for (int k = 0; k < count; k++)                     // slot k + 1 per pad (slot-indexed like the reference's texture cache)
    if ((rc = upload_frame (h->st, (size_t) k + 1, &pads[k].frame, &dpads[k].frame))) return rc;
  VfHipFrame dout;
  if ((rc = output_frame (h->st, 0, &h->out, out, &dout))) return rc;
  if (count > 0) VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = comp_launch (h, dpads.data (), count, background, &dout, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  return download_frame (h->st, 0, &dout, out);
}

// pipelined host path: flight k uses staging slot k * COMP_FLIGHT_SLOTS for the output and the next `count` for the pads
// (the synchronous entry point uses flight 0's slots: it refuses to run while frames are in flight)
enum { COMP_FLIGHT_SLOTS = 65 };

int vfhip_compositor_submit (VfHipCompositor *h, const VfHipPadInput *pads, int count, int background, VfHipFrame *out)
{
  int rc = comp_check (h, pads, count, background, out);
  if (rc) return rc;
  if (count >= COMP_FLIGHT_SLOTS) return set_error (VFHIP_ERR_UNSUPPORTED, "at most %d pads per pipelined composite", COMP_FLIGHT_SLOTS - 1);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  Flights &fl = h->fl;
  if (fl.count >= 2) return set_error (VFHIP_ERR_INVALID, "two frames are already in flight: call vfhip_compositor_wait first");
  const int k = (fl.head + fl.count) & 1;
  const size_t base = (size_t) k * COMP_FLIGHT_SLOTS;
  std::vector<VfHipPadInput> dpads (pads, pads + count);
  for (int i = 0; i < count; i++)
    if ((rc = upload_frame (h->st, base + 1 + (size_t) i, &pads[i].frame, &dpads[i].frame))) return rc;
  VfHipFrame dout;
  if ((rc = output_frame (h->st, base, &h->out, out, &dout))) return rc;
  if (count > 0) VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if ((rc = comp_launch (h, dpads.data (), count, background, &dout, h->st.s_compute))) return rc;
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  fl.f[k].out = *out;
  if ((rc = download_begin (h->st, base, &fl.f[k].out, fl.f[k].staged, h->st.ev_done[k]))) return rc;
  fl.count++;
  return VFHIP_OK;
}

int vfhip_compositor_wait (VfHipCompositor *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  Flights &fl = h->fl;
  if (fl.count == 0) return set_error (VFHIP_ERR_INVALID, "no frame in flight");
  const int k = fl.head;
  fl.head ^= 1; fl.count--;
  return download_finish (h->st, (size_t) k * COMP_FLIGHT_SLOTS, &fl.f[k].out, fl.f[k].staged, h->st.ev_done[k]);
}

int vfhip_compositor_in_flight (VfHipCompositor *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

void vfhip_compositor_cleanup (VfHipCompositor *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  for (int k = 0; k < 2; k++) { if (h->scratch[k]) (void) hipFree (h->scratch[k]); h->scratch[k] = nullptr; }
  for (auto &b : h->st.slots) { if (b.host) (void) hipHostFree (b.host); if (b.devp) (void) hipFree (b.devp); }
  h->st.slots.clear ();
  h->configured = false;
}

void vfhip_compositor_free (VfHipCompositor *h)
{
  if (!h) return;
  vfhip_compositor_cleanup (h);
  h->st.destroy ();
  delete h;
}

}  // extern "C"
