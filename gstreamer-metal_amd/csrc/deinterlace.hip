// csrc/deinterlace.hip — vfhip_deinterlace_* : bob / weave / linear / greedy-H deinterlacer.
// Mirrors MetalDeinterlaceRenderer (reference deinterlace/metaldeinterlacerenderer.{h,m}) and restates the
// kernels of deinterlace/metaldeinterlace_shaders.h:45-218 (`metal` numerics: float on unorm8, 8-bit RGBA
// intermediates).  What the reference does in 4-5 passes over 8-bit RGBA textures (YUV->RGBA render pass, the
// deinterlace compute pass, RGBA->YUV compute pass, a blit for the history and two command-buffer waits,
// metaldeinterlacerenderer.m:295-413) is ONE kernel here: the 8-bit RGBA intermediate of a pixel is a pure
// function of the input bytes, so it is recomputed in registers for the (at most three) taps a pixel needs,
// and the history is the previous INPUT frame in its native format (12.4 MB for NV12 2160p instead of 33 MB).
#include "vfhip_internal.h"
#include "metal_common.h"
#include <cmath>
#include <cstdlib>

using namespace vfhip;

namespace vfhip {

// k_deinterlace_420q's pixel pairs: two scalars (a packed-f32 form of the same expressions, v_pk_mul / add / fma_f32, gave the same bytes 4 % slower:
// the kernel's header; git history has it)
struct f2 { float x, y; };
__device__ __forceinline__ f2 operator+ (f2 a, f2 b) { return f2 { a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ f2 operator- (f2 a, f2 b) { return f2 { a.x - b.x, a.y - b.y }; }
__device__ __forceinline__ f2 operator* (f2 a, f2 b) { return f2 { a.x * b.x, a.y * b.y }; }
__device__ __forceinline__ f2 operator* (f2 a, float b) { return f2 { a.x * b, a.y * b }; }
__device__ __forceinline__ f2 operator* (float a, f2 b) { return f2 { a * b.x, a * b.y }; }
__device__ __forceinline__ f2 operator+ (f2 a, float b) { return f2 { a.x + b, a.y + b }; }
__device__ __forceinline__ f2 operator- (f2 a, float b) { return f2 { a.x - b, a.y - b }; }

struct DeintParams {
  metal::Img cur, prev;      // prev.p[0] == nullptr: no history
  metal::OutImg out;
  int method, tff;
  float threshold;
  // batch of consecutive frames of ONE stream: frame z at base + z * pitch; its history is frame z-1 of the batch
  // (frame 0: `prev`, the handle's stored history)
  size_t in_pitch, out_pitch;
};

__device__ __forceinline__ DeintParams deint_frame (const DeintParams &p, unsigned z)
{
  DeintParams q = p;
  q.cur = metal::img_at (p.cur, z * p.in_pitch);
  q.out = metal::out_at (p.out, z * p.out_pitch);
  if (z > 0) q.prev = metal::img_at (p.cur, (z - 1) * p.in_pitch);
  return q;
}

// the reference's _inputRGBA texel: YUV -> RGB with NEAREST chroma, quantised to 8 bits; RGBA bytes pass through
__device__ __forceinline__ uint32_t deint_input_rgba8 (const metal::Img &im, int x, int y)
{
  return metal::quant_rgba8 (metal::fetch_1to1 (im, x, y, false));
}

__device__ __forceinline__ uint32_t deint_pixel (const DeintParams &p, int x, int y)
{
  const int h = p.out.h;
  const bool top = (y & 1) == 0;
  const bool keep = p.tff ? top : !top;
  const uint32_t c = deint_input_rgba8 (p.cur, x, y);
  if (keep) return c;
  int method = p.method;
  if ((method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH) && !p.prev.p[0]) method = VFHIP_DEINTERLACE_BOB;
  if (method == VFHIP_DEINTERLACE_WEAVE) return deint_input_rgba8 (p.prev, x, y);
  bool bob = true;
  uint32_t pq = 0;
  if (method == VFHIP_DEINTERLACE_GREEDYH) {
    pq = deint_input_rgba8 (p.prev, x, y);
    const metal::F4 cl = metal::unpack_rgba8 (c), pl = metal::unpack_rgba8 (pq);
    const float dr = cl.r - pl.r, dg = cl.g - pl.g, db = cl.b - pl.b;
    const float motion = sqrtf (dr * dr + dg * dg + db * db);
    bob = !(motion < p.threshold);
  }
  if (!bob) return pq;
  const int above = y > 0 ? y - 1 : 0, below = y < h - 1 ? y + 1 : h - 1;
  const metal::F4 a = metal::unpack_rgba8 (deint_input_rgba8 (p.cur, x, above)), b = metal::unpack_rgba8 (deint_input_rgba8 (p.cur, x, below));
  metal::F4 o;
  o.r = (a.r + b.r) * 0.5f; o.g = (a.g + b.g) * 0.5f; o.b = (a.b + b.b) * 0.5f; o.a = (a.a + b.a) * 0.5f;
  return metal::quant_rgba8 (o);
}

__global__ __launch_bounds__ (256) void k_deinterlace (const DeintParams pp)
{
  const DeintParams p = deint_frame (pp, blockIdx.z);
  const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
  if (2 * bx >= p.out.w || 2 * by >= p.out.h) return;
  uint32_t q[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++)
      q[dy][dx] = deint_pixel (p, min (2 * bx + dx, p.out.w - 1), min (2 * by + dy, p.out.h - 1));
  metal::store_block (p.out, bx, by, q);
}

// ---- 4:2:0 inputs: column strips with a sliding window -------------------------------------------------------
// One lane owns one chroma column (2 pixels wide) and walks down a strip of rows two at a time.  The 8-bit RGBA
// intermediate of each source row is computed ONCE and carried in registers to serve as the "above" / "below" tap of
// its neighbours (k_deinterlace recomputes it up to three times); the previous frame is only converted on the lines
// that need it.  Same arithmetic, same results; ~1.5 instead of ~2.5 YUV->RGB conversions per pixel.
struct Q2 { uint32_t a, b; };          // logical RGBA8 of pixels (2k, y) and (2k+1, y)

template <bool PLANAR>
__device__ __forceinline__ Q2 deint_row_rgba8 (const metal::Img &im, int k, int y)
{
  const int x0 = 2 * k, x1 = min (2 * k + 1, im.w - 1);
  const uint8_t *yr = im.p[0] + (size_t) y * im.s[0];
  const int cy = y >> 1;
  typedef uint16_t __attribute__ ((aligned (1))) u16_any;        // 2-byte loads at any address (one instruction instead of two)
  const uint32_t yy = (x1 != x0) ? (uint32_t) *reinterpret_cast<const u16_any *> (yr + x0) : (uint32_t) yr[x0] * 0x0101u;
  const uint32_t Y0 = yy & 0xffu, Y1 = yy >> 8;
  uint32_t U, V;
  if (PLANAR) { U = im.p[1][(size_t) cy * im.s[1] + k]; V = im.p[2][(size_t) cy * im.s[2] + k]; }
  else { const uint32_t c = *reinterpret_cast<const u16_any *> (im.p[1] + (size_t) cy * im.s[1] + 2 * k); U = c & 0xffu; V = c >> 8; }
  const float cb = metal::un8 (U), cr = metal::un8 (V);
  Q2 q;
  q.a = metal::quant_rgba8 (metal::yuv_to_rgb (metal::un8 (Y0), cb, cr, im.m709));
  q.b = metal::quant_rgba8 (metal::yuv_to_rgb (metal::un8 (Y1), cb, cr, im.m709));
  return q;
}

__device__ __forceinline__ uint32_t deint_bob8 (uint32_t above, uint32_t below)
{
  const metal::F4 a = metal::unpack_rgba8 (above), b = metal::unpack_rgba8 (below);
  metal::F4 o;
  o.r = (a.r + b.r) * 0.5f; o.g = (a.g + b.g) * 0.5f; o.b = (a.b + b.b) * 0.5f; o.a = (a.a + b.a) * 0.5f;
  return metal::quant_rgba8 (o);
}

// output pixel of a discarded-field line given its own, above, below and previous-frame RGBA8
__device__ __forceinline__ uint32_t deint_other8 (int method, float thr, uint32_t cur, uint32_t above, uint32_t below, uint32_t prev)
{
  if (method == VFHIP_DEINTERLACE_WEAVE) return prev;
  if (method == VFHIP_DEINTERLACE_GREEDYH) {
    const metal::F4 cl = metal::unpack_rgba8 (cur), pl = metal::unpack_rgba8 (prev);
    const float dr = cl.r - pl.r, dg = cl.g - pl.g, db = cl.b - pl.b;
    if (sqrtf (dr * dr + dg * dg + db * db) < thr) return prev;
  }
  return deint_bob8 (above, below);
}

constexpr int DEINT_ROWS = 8;

template <bool PLANAR>
__global__ __launch_bounds__ (256) void k_deinterlace_420 (const DeintParams pp)
{
  const DeintParams p = deint_frame (pp, blockIdx.y);
  const int cw = (p.out.w + 1) >> 1, h = p.out.h;
  const int strips = (h + DEINT_ROWS - 1) / DEINT_ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= cw * strips) return;
  const int strip = t / cw, k = t - strip * cw;
  const int y0 = strip * DEINT_ROWS, yend = min (y0 + DEINT_ROWS, h);
  int method = p.method;
  if ((method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH) && !p.prev.p[0]) method = VFHIP_DEINTERLACE_BOB;
  const bool need_prev = method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH;
  Q2 qa = deint_row_rgba8<PLANAR> (p.cur, k, max (y0 - 1, 0));      // row above the pair
  Q2 qb = deint_row_rgba8<PLANAR> (p.cur, k, y0);                    // first row of the pair
  for (int y = y0; y < yend; y += 2) {
    const int r1 = min (y + 1, h - 1), r2 = min (y + 2, h - 1);
    const Q2 qc = deint_row_rgba8<PLANAR> (p.cur, k, r1);            // second row of the pair
    const Q2 qd = deint_row_rgba8<PLANAR> (p.cur, k, r2);            // row below the pair (= first row of the next pair)
    const bool keep0 = p.tff ? true : false;                          // y is even: top field
    uint32_t q[2][2];
    if (keep0) {                                                       // row y kept, row y+1 reconstructed
      q[0][0] = qb.a; q[0][1] = qb.b;
      Q2 pv = { 0u, 0u };
      if (need_prev) pv = deint_row_rgba8<PLANAR> (p.prev, k, r1);
      q[1][0] = deint_other8 (method, p.threshold, qc.a, qb.a, qd.a, pv.a);
      q[1][1] = deint_other8 (method, p.threshold, qc.b, qb.b, qd.b, pv.b);
    } else {                                                           // row y reconstructed (above = y-1 clamped), row y+1 kept
      Q2 pv = { 0u, 0u };
      if (need_prev) pv = deint_row_rgba8<PLANAR> (p.prev, k, y);
      q[0][0] = deint_other8 (method, p.threshold, qb.a, qa.a, qc.a, pv.a);
      q[0][1] = deint_other8 (method, p.threshold, qb.b, qa.b, qc.b, pv.b);
      q[1][0] = qc.a; q[1][1] = qc.b;
    }
    if (y + 1 >= h) { q[1][0] = q[0][0]; q[1][1] = q[0][1]; }         // odd height: edge-clamped duplicate for the 2x2 mean
    if (2 * k + 1 >= p.out.w) { q[0][1] = q[0][0]; q[1][1] = q[1][0]; }
    metal::store_block (p.out, k, y >> 1, q);
    qa = qc; qb = qd;
  }
}

// ---- 4:2:0 in and out, width % 4 == 0, even height: four pixels per lane, float intermediates ------------------------
// k_deinterlace_420q.  Same values as the kernels above (and as the reference's three passes), organised for the
// machine: a lane owns FOUR adjacent pixels (one dword of luma, two chroma columns) and walks a strip of rows in pairs.
//   * The reference's 8-bit RGBA intermediate lives in registers as the float an 8-bit texel reads back as — byte / 255
//     with the byte obtained by x255, round-to-nearest-even (`quant_sat2`) — so nothing is packed to bytes and unpacked again
//     between the input pass, the method pass and the RGB -> YUV pass (each row used to be unpacked up to three times).
//   * Every source row is converted once per strip and carried as the above / below tap of its neighbours; the chroma of a
//     row pair is converted once (the two rows of a 4:2:0 pair read the same chroma row).
//   * The four pixels are held as the PAIRS (0, 2) and (1, 3): both pairs take the lane's two chroma samples (u0, u1) as they are, and the 2x2
//     chroma means of the output come out of the same adds in the reference's summation order.  Each float operation is the one the reference's
//     passes make, on the same operands; three products by powers of two are folded where that is exact: (a + b) * 0.5 * 255 == (a + b) * 127.5
//     and (k * (s * 0.25)) + c == fma (k * s, 0.25, c) (scaling by 2^-n commutes with rounding away from the denormals).
//   * Scalar f32, not packed: a pair evaluated with v_pk_mul / add / fma_f32 gives the same bytes with a third fewer instructions and is 4 %
//     SLOWER.  The conversions and roundings in the stream (v_cvt_f32_ubyte, v_rndne_f32: 80 of ~400 per row pair) cost 3.4 cycles next to packed
//     instructions but ~2 when they alternate with scalar full-rate ones — they overlap (tools/ubench/valu_mix.hip, profiles/r03ah_valu_mix.txt) —
//     and fp32 peak is the same for both forms (64 flop per cycle and SIMD).
//   * The matrix is a template parameter, its coefficients LITERALS: v_fmamk / v_fmaak_f32 (a VOP2 multiply-add with a literal, 1.9 cycles) where a
//     coefficient in a register made it v_fma_f32 (VOP3: 3.5 cycles; in an SGPR 3.9) and cost a VGPR per coefficient.  The quad kernel therefore
//     wants input, history and output on one matrix (the element's frames are); anything else keeps k_deinterlace_420.
//   * clamp01 (x) ahead of the 8-bit quantisation is not an instruction of its own, nor a modifier on the (then VOP3) multiply-add: the clamp sits
//     on the LAST multiply, rint (x * 255) * (1 / 255) -> [0, 1].  Equal for every x: inside [0, 1] the clamp does nothing on either side
//     (255 * fl (1 / 255) rounds to exactly 1); above, both give 1; below, both give 0 (rint and the products are monotone).
//   * greedy-H compares the squared distance with the smallest float whose correctly rounded square root reaches the
//     threshold (computed on the host: `motion2_limit`), which decides exactly like sqrt (d2) < threshold without the
//     square root.
//   * Dword loads / stores on luma and on NV12 chroma, 32-bit offsets from wave-uniform plane bases.
struct Rgb4 { f2 r[2], g[2], b[2]; };                    // [0]: pixels 0 and 2 of the lane's four, [1]: pixels 1 and 3
struct Chroma2 { f2 u, v; };                             // cb - 128/255, cr - 128/255 of the lane's two chroma columns
// the two matrices of metal_common.h (yuv_to_rgb / rgb_to_yuv) as compile-time coefficient sets
template <bool M709> struct YuvK {
  static constexpr float rv = M709 ? 1.792741f : 1.596027f, gu = M709 ? -0.213249f : -0.391762f, gv = M709 ? -0.532909f : -0.812968f, bu = M709 ? 2.112402f : 2.017232f;
};
template <bool M709> struct RgbK {
  static constexpr float yr = M709 ? 0.182586f : 0.256788f, yg = M709 ? 0.614231f : 0.504129f, yb = M709 ? 0.062007f : 0.097906f;
  static constexpr float ur = M709 ? -0.100644f : -0.148223f, ug = M709 ? -0.338572f : -0.290993f, ub = 0.439216f;
  static constexpr float vr = 0.439216f, vg = M709 ? -0.398942f : -0.367788f, vb = M709 ? -0.040274f : -0.071427f;
};

__device__ __forceinline__ f2 rint2 (f2 x) { return f2 { __builtin_rintf (x.x), __builtin_rintf (x.y) }; }
// k * a + c with a literal k
__device__ __forceinline__ f2 fmak2 (float k, f2 a, f2 c) { return f2 { fmaf (k, a.x, c.x), fmaf (k, a.y, c.y) }; }
// what an 8-bit unorm texel written with clamp01 (x) reads back as: the clamp on the last multiply (header).  r255: 1 / 255 in a VGPR (a VOP3
// instruction takes no literal, and an SGPR operand would halve its rate)
__device__ __forceinline__ float mul_sat (float a, float b) { float d; asm ("v_mul_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f2 quant_sat2 (f2 x, float r255)
{
  const f2 q = rint2 (x * 255.0f);
  return f2 { mul_sat (q.x, r255), mul_sat (q.y, r255) };
}
__device__ __forceinline__ float dq_vgpr (float x) { float v; asm volatile ("v_mov_b32 %0, %1" : "=v"(v) : "s"(x)); return v; }

// raw loads: one dword of luma (four pixels), the lane's two chroma columns of row y >> 1 as one dword
// (NV12: U0 V0 U1 V1; planar: U0 U1 V0 V1)
__device__ __forceinline__ uint32_t deint_luma4 (const uint8_t *yp, uint32_t ys, uint32_t q, int y)
{
  return *reinterpret_cast<const uint32_t *> (yp + (__umul24 ((uint32_t) y, ys) + 4u * q));
}
template <bool PLANAR>
__device__ __forceinline__ uint32_t deint_chroma4 (const uint8_t *up, const uint8_t *vp, uint32_t cs, uint32_t q, int y)
{
  if (PLANAR) {
    const uint32_t co = __umul24 ((uint32_t) (y >> 1), cs) + 2u * q;
    return (uint32_t) *reinterpret_cast<const uint16_t *> (up + co) | ((uint32_t) *reinterpret_cast<const uint16_t *> (vp + co) << 16);
  }
  return *reinterpret_cast<const uint32_t *> (up + (__umul24 ((uint32_t) (y >> 1), cs) + 4u * q));
}
template <bool PLANAR>
__device__ __forceinline__ Chroma2 deint_chroma2 (uint32_t c4)
{
  f2 cb, cr;
  if (PLANAR) { cb = f2 { (float) (c4 & 0xffu), (float) ((c4 >> 8) & 0xffu) }; cr = f2 { (float) ((c4 >> 16) & 0xffu), (float) (c4 >> 24) }; }
  else { cb = f2 { (float) (c4 & 0xffu), (float) ((c4 >> 16) & 0xffu) }; cr = f2 { (float) ((c4 >> 8) & 0xffu), (float) (c4 >> 24) }; }
  Chroma2 c;
  c.u = cb * (1.0f / 255.0f) - 128.0f / 255.0f; c.v = cr * (1.0f / 255.0f) - 128.0f / 255.0f;      // metal::un8, then yuv_to_rgb's offsets
  return c;
}

template <bool M709>
__device__ __forceinline__ Rgb4 deint_row4 (uint32_t Y4, const Chroma2 &c, float r255)
{
  typedef YuvK<M709> K;
  const f2 yb[2] = { f2 { (float) (Y4 & 0xffu), (float) ((Y4 >> 16) & 0xffu) }, f2 { (float) ((Y4 >> 8) & 0xffu), (float) (Y4 >> 24) } };
  Rgb4 o;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    // metal::yuv_to_rgb (same operations, same order)
    const f2 ly = 1.164383f * (yb[i] * (1.0f / 255.0f) - 16.0f / 255.0f);
    o.r[i] = quant_sat2 (fmak2 (K::rv, c.v, ly), r255);
    o.g[i] = quant_sat2 (fmak2 (K::gv, c.v, fmak2 (K::gu, c.u, ly)), r255);
    o.b[i] = quant_sat2 (fmak2 (K::bu, c.u, ly), r255);
  }
  return o;
}

// rows y (o0) and y+1 (o1) of one lane's four columns -> NV12 / I420 (the reference's rgbaToNV12 / rgbaToI420 pass:
// luma per pixel, chroma from the mean of each 2x2 block, summed in the reference's order)
template <bool PLANAR, bool M709>
__device__ __forceinline__ void deint_store4 (const metal::OutImg &o, uint32_t q, int y, const Rgb4 &o0, const Rgb4 &o1)
{
  typedef RgbK<M709> K;
  uint32_t l0 = 0, l1 = 0;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    // metal::rgb_to_yuv (luma row); pair i holds the bytes i and i + 2 of the dword
    const f2 Y0 = (fmak2 (K::yb, o0.b[i], fmak2 (K::yg, o0.g[i], K::yr * o0.r[i])) + 16.0f / 255.0f) * 255.0f;
    const f2 Y1 = (fmak2 (K::yb, o1.b[i], fmak2 (K::yg, o1.g[i], K::yr * o1.r[i])) + 16.0f / 255.0f) * 255.0f;
    l0 = __builtin_amdgcn_cvt_pk_u8_f32 (Y0.x, (uint32_t) i, l0); l0 = __builtin_amdgcn_cvt_pk_u8_f32 (Y0.y, (uint32_t) i + 2u, l0);
    l1 = __builtin_amdgcn_cvt_pk_u8_f32 (Y1.x, (uint32_t) i, l1); l1 = __builtin_amdgcn_cvt_pk_u8_f32 (Y1.y, (uint32_t) i + 2u, l1);
  }
  const uint32_t lo = __umul24 ((uint32_t) y, (uint32_t) o.s[0]) + 4u * q;
  __builtin_nontemporal_store (l0, reinterpret_cast<uint32_t *> (o.p[0] + lo));
  __builtin_nontemporal_store (l1, reinterpret_cast<uint32_t *> (o.p[0] + (lo + (uint32_t) o.s[0])));
  // the two 2x2 blocks at once: ((row0 left + row0 right) + row1 left) + row1 right, x 0.25 folded into the last fma
  const f2 sr = ((o0.r[0] + o0.r[1]) + o1.r[0]) + o1.r[1], sg = ((o0.g[0] + o0.g[1]) + o1.g[0]) + o1.g[1], sb = ((o0.b[0] + o0.b[1]) + o1.b[0]) + o1.b[1];
  const f2 c128 = f2 { 128.0f / 255.0f, 128.0f / 255.0f };
  const f2 U = fmak2 (0.25f, fmak2 (K::ub, sb, fmak2 (K::ug, sg, K::ur * sr)), c128) * 255.0f;
  const f2 V = fmak2 (0.25f, fmak2 (K::vb, sb, fmak2 (K::vg, sg, K::vr * sr)), c128) * 255.0f;
  if (PLANAR) {
    uint32_t uu = __builtin_amdgcn_cvt_pk_u8_f32 (U.x, 0u, 0u), vv = __builtin_amdgcn_cvt_pk_u8_f32 (V.x, 0u, 0u);
    uu = __builtin_amdgcn_cvt_pk_u8_f32 (U.y, 1u, uu); vv = __builtin_amdgcn_cvt_pk_u8_f32 (V.y, 1u, vv);
    const uint32_t co = __umul24 ((uint32_t) (y >> 1), (uint32_t) o.s[1]) + 2u * q;
    *reinterpret_cast<uint16_t *> (o.p[1] + co) = (uint16_t) uu;
    *reinterpret_cast<uint16_t *> (o.p[2] + (__umul24 ((uint32_t) (y >> 1), (uint32_t) o.s[2]) + 2u * q)) = (uint16_t) vv;
  } else {
    uint32_t c4 = __builtin_amdgcn_cvt_pk_u8_f32 (U.x, 0u, 0u);
    c4 = __builtin_amdgcn_cvt_pk_u8_f32 (V.x, 1u, c4); c4 = __builtin_amdgcn_cvt_pk_u8_f32 (U.y, 2u, c4); c4 = __builtin_amdgcn_cvt_pk_u8_f32 (V.y, 3u, c4);
    const uint32_t co = __umul24 ((uint32_t) (y >> 1), (uint32_t) o.s[1]) + 4u * q;
    __builtin_nontemporal_store (c4, reinterpret_cast<uint32_t *> (o.p[1] + co));
  }
}

// greedy-H's motion test of the lane's four pixels (bit k: pixel k of the pairs' order 0, 2, 1, 3 has NOT moved against the
// previous frame: weave it)
struct Still4 { bool s[2][2]; };
__device__ __forceinline__ Still4 deint_still4 (const Rgb4 &cur, const Rgb4 &prev, float m2_limit)
{
  Still4 st;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const f2 dr = cur.r[i] - prev.r[i], dg = cur.g[i] - prev.g[i], db = cur.b[i] - prev.b[i];
    const f2 d2 = (dr * dr + dg * dg) + db * db;
    st.s[i][0] = d2.x < m2_limit; st.s[i][1] = d2.y < m2_limit;
  }
  return st;
}
// the reconstructed line from its neighbours in the kept field (bob) and the previous frame's pixel (weave / still pixels of greedy-H)
template <int METHOD>
__device__ __forceinline__ Rgb4 deint_recon4 (const Still4 &st, const Rgb4 &above, const Rgb4 &below, const Rgb4 &prev)
{
  if (METHOD == VFHIP_DEINTERLACE_WEAVE) return prev;
  Rgb4 o;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    // the mean of two texel values is inside [0, 1]: no clamp.  (a + b) * 0.5 * 255 == (a + b) * 127.5 exactly.
    const f2 br = rint2 ((above.r[i] + below.r[i]) * 127.5f) * (1.0f / 255.0f), bg = rint2 ((above.g[i] + below.g[i]) * 127.5f) * (1.0f / 255.0f),
             bb = rint2 ((above.b[i] + below.b[i]) * 127.5f) * (1.0f / 255.0f);
    if (METHOD == VFHIP_DEINTERLACE_GREEDYH) {      // selects, no divergent branch
      const bool s0 = st.s[i][0], s1 = st.s[i][1];
      o.r[i] = f2 { s0 ? prev.r[i].x : br.x, s1 ? prev.r[i].y : br.y };
      o.g[i] = f2 { s0 ? prev.g[i].x : bg.x, s1 ? prev.g[i].y : bg.y };
      o.b[i] = f2 { s0 ? prev.b[i].x : bb.x, s1 ? prev.b[i].y : bb.y };
    } else {
      o.r[i] = br; o.g[i] = bg; o.b[i] = bb;
    }
  }
  return o;
}

constexpr int DEINTQ_ROWS = 8;      // rows per lane for one frame (batches: 16 / 32, deint_launch)

// one lane's strip.  HIST (wave-uniform, decided outside the row loop so that the loop body is straight-line code): false = frame 0
// of a stream (or of a batch on a fresh handle), which has no history — weave / greedy-H fall back to bob for that frame only,
// the rest of the batch uses its predecessor in the batch.
// The loop is software-pipelined by hand: the (up to five) dwords a row pair needs are loaded one iteration ahead, right after
// the previous pair's have been converted, so that their latency is covered by this lane's own arithmetic and not only by the
// other waves of the SIMD (rows past the frame are clamped: a load too many per strip, never a branch).
template <bool PLANAR, bool TFF, int METHOD, bool HIST, bool M709>
__device__ __forceinline__ void deint_strip (const DeintParams &p, uint32_t q, int y0, int yend, float m2_limit)
{
  constexpr int M = (METHOD == VFHIP_DEINTERLACE_WEAVE || METHOD == VFHIP_DEINTERLACE_GREEDYH) && !HIST ? VFHIP_DEINTERLACE_BOB : METHOD;
  constexpr bool NEED_PREV = M == VFHIP_DEINTERLACE_WEAVE || M == VFHIP_DEINTERLACE_GREEDYH;
  constexpr bool GREEDY = M == VFHIP_DEINTERLACE_GREEDYH;
  const int h = p.out.h;
  const uint8_t *cy = p.cur.p[0], *cu = p.cur.p[1], *cv = p.cur.p[2];
  const uint8_t *py = p.prev.p[0], *pu = p.prev.p[1], *pv = p.prev.p[2];
  const uint32_t ys = (uint32_t) p.cur.s[0], cs = (uint32_t) p.cur.s[1], pys = (uint32_t) p.prev.s[0], pcs = (uint32_t) p.prev.s[1];
  const float r255 = dq_vgpr (1.0f / 255.0f);
  // TFF: even rows are kept, odd rows reconstructed from the kept rows above (y) and below (y + 2);
  // BFF: odd rows are kept, even rows reconstructed from the kept rows above (y - 1) and below (y + 1).
  // y0 and y are even: rows y and y + 1 share chroma row y >> 1.
  // per row pair: `t` = the reconstructed row itself (only greedy-H's motion test reads it), `k` = the kept row that enters the
  // pair new (TFF: below, y + 2; BFF: y + 1), `pr` = row t of the previous frame; chroma comes with k (TFF) or t (BFF) and with pr
  uint32_t y_t = 0, y_k = 0, c_k = 0, y_p = 0, c_p = 0;
  auto fetch = [&] (int y) {
    const int rt = min (TFF ? y + 1 : y, h - 1), rk = min (TFF ? y + 2 : y + 1, h - 1);
    if (GREEDY) y_t = deint_luma4 (cy, ys, q, rt);
    y_k = deint_luma4 (cy, ys, q, rk);
    c_k = deint_chroma4<PLANAR> (cu, cv, cs, q, rk);
    if (NEED_PREV) { y_p = deint_luma4 (py, pys, q, rt); c_p = deint_chroma4<PLANAR> (pu, pv, pcs, q, rt); }
  };
  const int yc = TFF ? y0 : max (y0 - 1, 0);
  const uint32_t y_c = deint_luma4 (cy, ys, q, yc), c_c = deint_chroma4<PLANAR> (cu, cv, cs, q, yc);
  fetch (y0);
  Chroma2 cc = deint_chroma2<PLANAR> (c_c);
  Rgb4 ra = deint_row4<M709> (y_c, cc, r255);
  // one row pair: `above` = the kept row over the reconstructed one (it came in with the pair before), `kept` = the one this pair brings
  auto pair = [&] (int y, const Rgb4 &above, Rgb4 &kept) {
    Rgb4 prev {};
    Still4 st {};
    // (the scheduling barriers keep the three row conversions from being interleaved: one at a time they fit in 80 VGPRs, six waves per SIMD; without them: the same speed)
    if (NEED_PREV) prev = deint_row4<M709> (y_p, deint_chroma2<PLANAR> (c_p), r255);
    __builtin_amdgcn_sched_barrier (0);
    if (TFF) {
      if (GREEDY) st = deint_still4 (deint_row4<M709> (y_t, cc, r255), prev, m2_limit);          // row y + 1 shares the chroma of row y
      cc = deint_chroma2<PLANAR> (c_k);
    } else {
      cc = deint_chroma2<PLANAR> (c_k);
      if (GREEDY) st = deint_still4 (deint_row4<M709> (y_t, cc, r255), prev, m2_limit);
    }
    __builtin_amdgcn_sched_barrier (0);
    kept = deint_row4<M709> (y_k, cc, r255);
    __builtin_amdgcn_sched_barrier (0);
    fetch (y + 2);
    const Rgb4 rec = deint_recon4<M> (st, above, kept, prev);
    __builtin_amdgcn_sched_barrier (0);
    if (TFF) deint_store4<PLANAR, M709> (p.out, q, y, above, rec);
    else deint_store4<PLANAR, M709> (p.out, q, y, rec, kept);
  };
  // two pairs per trip: the kept row changes hands between `ra` and `rb` instead of being copied (six 64-bit moves per pair)
  Rgb4 rb;
  for (int y = y0; y < yend; y += 4) {
    pair (y, ra, rb);
    if (y + 2 >= yend) break;
    pair (y + 2, rb, ra);
  }
}

template <bool PLANAR, bool TFF, int METHOD, bool M709>
__global__ __launch_bounds__ (256) void k_deinterlace_420q (const DeintParams pp, float m2_limit, int rows)
{
  const DeintParams p = deint_frame (pp, blockIdx.y);
  const int quads = p.out.w >> 2, h = p.out.h;
  const int strips = (h + rows - 1) / rows;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= quads * strips) return;
  const int strip = t / quads;
  const uint32_t q = (uint32_t) (t - strip * quads);
  const int y0 = strip * rows, yend = min (y0 + rows, h);
  if (METHOD == VFHIP_DEINTERLACE_BOB || p.prev.p[0] != nullptr) deint_strip<PLANAR, TFF, METHOD, true, M709> (p, q, y0, yend, m2_limit);
  else deint_strip<PLANAR, TFF, METHOD, false, M709> (p, q, y0, yend, m2_limit);
}

}  // namespace vfhip

struct VfHipDeinterlace {
  std::mutex mu;
  Device *dev = nullptr;
  Staging st;
  bool configured = false;
  VfHipVideoInfo info {};
  // host path: three upload slots used in turn (the previous one is the history; with two frames in flight the third is
  // the only one no kernel can still be reading), outputs in slots 3 / 4
  unsigned seq = 0;
  Flights fl;                       // pipelined host path (submit / wait)
  bool has_prev = false;
  VfHipFrame prev_dev {};           // device-side previous input frame (host path: staging slot; device path: hist buffer)
  void *hist[2] = { nullptr, nullptr }; size_t hist_bytes = 0;   // device path: ping-pong history images
  int hist_cur = 0;
};

// smallest float d with sqrtf (d) >= thr: (sqrtf (m2) < thr) == (m2 < d) for every m2 >= 0, because the correctly rounded
// square root is monotone.  thr <= 0 -> 0 (never below), NaN -> NaN (never below), beyond sqrt (FLT_MAX) -> +inf.
static float motion2_limit (float thr)
{
  if (!(thr > 0.0f)) return thr != thr ? thr : 0.0f;
  float d = thr * thr;
  if (std::isinf (d)) return d;
  while (d > 0.0f && sqrtf (d) >= thr) d = nextafterf (d, 0.0f);
  while (sqrtf (d) < thr) d = nextafterf (d, INFINITY);
  return d;
}

// k_deinterlace_420q's contract: 4:2:0 frame, width % 4 == 0, even height, every plane (and the batch pitches) aligned
// for the dword / 16-bit accesses it makes, input / history / output on one colour matrix (a template parameter there)
static bool deint_quad_ok (const VfHipVideoInfo &info, const VfHipFrame *cur, const VfHipFrame *prev, const VfHipFrame *out, size_t in_pitch, size_t out_pitch)
{
  static const bool enabled = [] { const char *e = getenv ("VFHIP_DEINT_QUAD"); return !e || atoi (e) != 0; } ();     // test / A-B knob
  if (!enabled) return false;
  const bool nv12 = info.format == VFHIP_FORMAT_NV12, i420 = info.format == VFHIP_FORMAT_I420;
  if (!(nv12 || i420) || (info.width & 3) || (info.height & 1) || info.width < 4 || info.height < 2) return false;
  if ((in_pitch | out_pitch) & 3) return false;
  const bool m709 = cur->info.color_matrix == VFHIP_MATRIX_BT709;
  if ((out->info.color_matrix == VFHIP_MATRIX_BT709) != m709 || (prev && (prev->info.color_matrix == VFHIP_MATRIX_BT709) != m709)) return false;
  auto ok = [&] (const VfHipFrame *f) {
    if (!f) return true;
    if (((uintptr_t) f->data[0] | (uintptr_t) f->stride[0]) & 3) return false;
    const uintptr_t cm = nv12 ? 3 : 1;
    if (((uintptr_t) f->data[1] | (uintptr_t) f->stride[1]) & cm) return false;
    if (i420 && (((uintptr_t) f->data[2] | (uintptr_t) f->stride[2]) & 1)) return false;
    if (i420 && f->stride[1] != f->stride[2]) return false;       // one chroma offset serves U and V on the input side
    return true;
  };
  return ok (cur) && ok (prev) && ok (out);
}

static int deint_launch (VfHipDeinterlace *h, const VfHipFrame *cur, const VfHipFrame *prev, VfHipFrame *out,
    const VfHipDeinterlaceParams *prm, hipStream_t s, int n_frames = 1, size_t in_pitch = 0, size_t out_pitch = 0)
{
  DeintParams p {};
  p.in_pitch = in_pitch; p.out_pitch = out_pitch;
  p.cur = metal::make_img (cur);
  if (prev) p.prev = metal::make_img (prev);
  p.out = metal::make_out (out);
  p.method = prm->method; p.tff = prm->top_field_first != 0; p.threshold = prm->motion_threshold;
  const int bw = (h->info.width + 1) / 2, bh = (h->info.height + 1) / 2;
  if (deint_quad_ok (h->info, cur, prev, out, in_pitch, out_pitch)) {
    // the whole frame in dwords: k_deinterlace_420q, one instantiation per (layout, field order, method)
    int method = p.method;
    if ((method == VFHIP_DEINTERLACE_WEAVE || method == VFHIP_DEINTERLACE_GREEDYH) && !p.prev.p[0] && n_frames == 1) method = VFHIP_DEINTERLACE_BOB;   // no history at all
    if (method == VFHIP_DEINTERLACE_LINEAR) method = VFHIP_DEINTERLACE_BOB;                    // the reference's linear IS bob (shaders.h:134-148)
    // rows per lane: a strip starts with one row conversion of its own (the row above it), so longer strips do less work, and a batch
    // that fills the chip many times over can afford half as many lanes (a single frame cannot: 1080p in 8-row strips is 1012 waves)
    static const int rows_env = [] { const char *e = getenv ("VFHIP_DEINT_ROWS"); return e ? atoi (e) : 0; } ();      // A/B knob (even, >= 2)
    const size_t lanes8 = (size_t) (h->info.width / 4) * ((h->info.height + DEINTQ_ROWS - 1) / DEINTQ_ROWS) * (size_t) n_frames;
    int rows = lanes8 >= ((size_t) 1 << 24) ? 4 * DEINTQ_ROWS : lanes8 >= ((size_t) 1 << 21) ? 2 * DEINTQ_ROWS : DEINTQ_ROWS;
    if (rows_env >= 2 && !(rows_env & 1)) rows = rows_env;
    const int strips = (h->info.height + rows - 1) / rows;
    dim3 grid ((unsigned) (((size_t) (h->info.width / 4) * strips + 255) / 256), (unsigned) n_frames);
    const float lim = motion2_limit (p.threshold);
    const bool planar = h->info.format == VFHIP_FORMAT_I420;
    const bool m709 = p.cur.m709 != 0;
#define VF_DQ(PL, TF, M) do { if (m709) hipLaunchKernelGGL ((k_deinterlace_420q<PL, TF, M, true>), grid, dim3 (256), 0, s, p, lim, rows); \
                              else hipLaunchKernelGGL ((k_deinterlace_420q<PL, TF, M, false>), grid, dim3 (256), 0, s, p, lim, rows); } while (0)
#define VF_DQ_M(PL, TF) do { if (method == VFHIP_DEINTERLACE_BOB) VF_DQ (PL, TF, VFHIP_DEINTERLACE_BOB); else if (method == VFHIP_DEINTERLACE_WEAVE) VF_DQ (PL, TF, VFHIP_DEINTERLACE_WEAVE); \
                              else VF_DQ (PL, TF, VFHIP_DEINTERLACE_GREEDYH); } while (0)
    if (planar) { if (p.tff) VF_DQ_M (true, true); else VF_DQ_M (true, false); }
    else { if (p.tff) VF_DQ_M (false, true); else VF_DQ_M (false, false); }
#undef VF_DQ_M
#undef VF_DQ
  } else if (h->info.format == VFHIP_FORMAT_NV12 || h->info.format == VFHIP_FORMAT_I420) {
    const int strips = (h->info.height + DEINT_ROWS - 1) / DEINT_ROWS;
    dim3 grid ((unsigned) (((size_t) bw * strips + 255) / 256), (unsigned) n_frames);
    if (h->info.format == VFHIP_FORMAT_I420) hipLaunchKernelGGL (k_deinterlace_420<true>, grid, dim3 (256), 0, s, p);
    else hipLaunchKernelGGL (k_deinterlace_420<false>, grid, dim3 (256), 0, s, p);
  } else {
    dim3 grid ((unsigned) ((bw + 63) / 64), (unsigned) ((bh + 3) / 4), (unsigned) n_frames);
    hipLaunchKernelGGL (k_deinterlace, grid, dim3 (64, 4), 0, s, p);
  }
  VFHIP_CHECK_HIP (hipGetLastError ());
  return VFHIP_OK;
}

static int deint_check (VfHipDeinterlace *h, const VfHipFrame *in, const VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  if (!h || !prm) return set_error (VFHIP_ERR_INVALID, "null argument");
  if (!h->configured) return set_error (VFHIP_ERR_NOT_CONFIGURED, "deinterlace: process before configure");
  if (prm->method < 0 || prm->method > 3) return set_error (VFHIP_ERR_INVALID, "bad deinterlace method %d", prm->method);
  int rc = check_frame (in, &h->info, "input");
  if (rc) return rc;
  return check_frame (out, &h->info, "output");
}

static int deint_device_locked (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_pitch, size_t out_pitch, int n_frames,
    const VfHipDeinterlaceParams *prm, hipStream_t s);

extern "C" {

VfHipDeinterlace *vfhip_deinterlace_new (int device)
{
  Device *d = get_device (device);
  if (!d) return nullptr;
  VfHipDeinterlace *h = new (std::nothrow) VfHipDeinterlace ();
  if (!h) { set_error (VFHIP_ERR_NOMEM, "out of memory"); return nullptr; }
  h->dev = d;
  if (h->st.init (d) != VFHIP_OK) { delete h; return nullptr; }
  return h;
}

int vfhip_deinterlace_configure (VfHipDeinterlace *h, const VfHipVideoInfo *info)
{
  if (!h || !info) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "configure with %d submitted frame(s) still in flight: wait for them first", h->fl.count);
  if (info->width <= 0 || info->height <= 0 || info->width > 32768 || info->height > 32768)
    return set_error (VFHIP_ERR_INVALID, "bad frame size %dx%d", info->width, info->height);
  // pad template of the reference: BGRA, RGBA, NV12, I420 (deinterlace/gstvfmetaldeinterlace.m:43-55)
  if (info->format < VFHIP_FORMAT_BGRA || info->format > VFHIP_FORMAT_I420)
    return set_error (VFHIP_ERR_UNSUPPORTED, "deinterlace: format %d not supported", info->format);
  h->info = *info;
  h->has_prev = false;               // reference resets the history on reconfigure (metaldeinterlacerenderer.m:180)
  h->configured = true;
  return VFHIP_OK;
}

int vfhip_deinterlace_reset (VfHipDeinterlace *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::lock_guard<std::mutex> lk (h->mu);
  h->has_prev = false;
  return VFHIP_OK;
}

// one frame onto the handle's streams: upload (or use in place) -> kernel with the history -> download queued behind it
static int deint_submit_locked (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  if (h->fl.count >= 2) return set_error (VFHIP_ERR_INVALID, "two frames are already in flight: call vfhip_deinterlace_wait first");
  const int k = (h->fl.head + h->fl.count) & 1;
  const size_t in_slot = h->seq % 3, out_slot = 3 + (size_t) k;
  VfHipFrame din, dout;
  int rc;
  if ((rc = upload_frame (h->st, in_slot, in, &din))) return rc;
  if ((rc = output_frame (h->st, out_slot, &h->info, out, &dout))) return rc;
  VFHIP_CHECK_HIP (hipStreamWaitEvent (h->st.s_compute, h->st.ev_h2d, 0));
  if (in->flags & VFHIP_FRAME_FLAG_DEVICE) {
    // device-resident input (memory:HIPMemory buffer): it belongs to the caller and may be recycled after this frame, so
    // the history is copied device-to-device like in the process_device path
    if ((rc = deint_device_locked (h, &din, &dout, 0, 0, 1, prm, h->st.s_compute))) return rc;
  } else {
    if ((rc = deint_launch (h, &din, h->has_prev ? &h->prev_dev : nullptr, &dout, prm, h->st.s_compute))) return rc;
    h->prev_dev = din; h->has_prev = true;              // the staging slot of this frame is the next frame's history
  }
  VFHIP_CHECK_HIP (hipEventRecord (h->st.ev_compute, h->st.s_compute));
  h->fl.f[k].out = *out;
  if ((rc = download_begin (h->st, out_slot, &h->fl.f[k].out, h->fl.f[k].staged, h->st.ev_done[k]))) return rc;
  h->fl.count++; h->seq++;
  return VFHIP_OK;
}

static int deint_wait_locked (VfHipDeinterlace *h)
{
  if (h->fl.count == 0) return set_error (VFHIP_ERR_INVALID, "no frame in flight");
  const int k = h->fl.head;
  h->fl.head ^= 1; h->fl.count--;
  return download_finish (h->st, 3 + (size_t) k, &h->fl.f[k].out, h->fl.f[k].staged, h->st.ev_done[k]);
}

int vfhip_deinterlace_process (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  int rc = deint_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  if (h->fl.count) return set_error (VFHIP_ERR_INVALID, "frames submitted with vfhip_deinterlace_submit are still in flight");
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  if ((rc = deint_submit_locked (h, in, out, prm))) return rc;
  return deint_wait_locked (h);
}

int vfhip_deinterlace_submit (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out, const VfHipDeinterlaceParams *prm)
{
  int rc = deint_check (h, in, out, prm);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return deint_submit_locked (h, in, out, prm);
}

int vfhip_deinterlace_wait (VfHipDeinterlace *h)
{
  if (!h) return set_error (VFHIP_ERR_INVALID, "null handle");
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return deint_wait_locked (h);
}

int vfhip_deinterlace_in_flight (VfHipDeinterlace *h)
{
  if (!h) return 0;
  std::lock_guard<std::mutex> lk (h->mu);
  return h->fl.count;
}

// history = the previous input frame, kept in one of two internal device images and filled by a stream-ordered
// device-to-device copy (the reference blits _inputRGBA -> _prevFrameRGBA, :394-405).  Writing the history from
// inside the kernel was measured 3x SLOWER (2-byte stores per lane: 9.2 k vs 27.3 k frames/s on NV12 2160p).
// In a batch the history of frame k is frame k-1 of the batch itself; only the LAST frame is copied.
static int deint_device_locked (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_pitch, size_t out_pitch, int n_frames,
    const VfHipDeinterlaceParams *prm, hipStream_t s)
{
  int rc;
  size_t total = 0, off[VFHIP_MAX_PLANES] = { 0 };
  const int np = format_n_planes (in0->info.format);
  for (int p = 0; p < np; p++) { off[p] = total; total += (frame_plane_bytes (in0, p) + 255) / 256 * 256; }
  if (h->hist_bytes < total) {
    for (int k = 0; k < 2; k++) { if (h->hist[k]) (void) hipFree (h->hist[k]); h->hist[k] = nullptr; }
    h->hist_bytes = 0; h->has_prev = false;
    for (int k = 0; k < 2; k++) VFHIP_CHECK_HIP (dev_malloc (&h->hist[k], total));
    h->hist_bytes = total;
  }
  VfHipFrame next = *in0;
  const int nxt = 1 - h->hist_cur;
  for (int p = 0; p < np; p++) next.data[p] = (uint8_t *) h->hist[nxt] + off[p];
  if ((rc = deint_launch (h, in0, h->has_prev ? &h->prev_dev : nullptr, out0, prm, s, n_frames, in_pitch, out_pitch))) return rc;
  const size_t last = (size_t) (n_frames - 1) * in_pitch;
  for (int p = 0; p < np; p++)
    VFHIP_CHECK_HIP (hipMemcpyAsync (next.data[p], (const uint8_t *) in0->data[p] + last, frame_plane_bytes (in0, p), hipMemcpyDeviceToDevice, s));
  h->prev_dev = next; h->hist_cur = nxt;
  h->has_prev = true;
  return VFHIP_OK;
}

static int deint_device (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0, size_t in_pitch, size_t out_pitch, int n_frames,
    const VfHipDeinterlaceParams *prm, void *stream)
{
  int rc = deint_check (h, in0, out0, prm);
  if (rc) return rc;
  if (n_frames < 1 || n_frames > 65535) return set_error (VFHIP_ERR_INVALID, "n_frames %d outside 1..65535", n_frames);
  std::lock_guard<std::mutex> lk (h->mu);
  VFHIP_CHECK_HIP (hipSetDevice (h->dev->ordinal));
  return deint_device_locked (h, in0, out0, in_pitch, out_pitch, n_frames, prm, stream ? (hipStream_t) stream : h->st.s_compute);
}

int vfhip_deinterlace_process_device (VfHipDeinterlace *h, const VfHipFrame *in, VfHipFrame *out,
    const VfHipDeinterlaceParams *prm, void *stream)
{
  return deint_device (h, in, out, 0, 0, 1, prm, stream);
}

int vfhip_deinterlace_process_device_batch (VfHipDeinterlace *h, const VfHipFrame *in0, VfHipFrame *out0,
    size_t in_frame_pitch, size_t out_frame_pitch, int n_frames, const VfHipDeinterlaceParams *prm, void *stream)
{
  return deint_device (h, in0, out0, in_frame_pitch, out_frame_pitch, n_frames, prm, stream);
}

void vfhip_deinterlace_cleanup (VfHipDeinterlace *h)
{
  if (!h) return;
  std::lock_guard<std::mutex> lk (h->mu);
  (void) hipSetDevice (h->dev->ordinal);
  flights_abandon (h->st, h->fl);
  for (int k = 0; k < 2; k++) { if (h->hist[k]) (void) hipFree (h->hist[k]); h->hist[k] = nullptr; }
  h->hist_bytes = 0;
  for (auto &b : h->st.slots) { if (b.host) (void) hipHostFree (b.host); if (b.devp) (void) hipFree (b.devp); }
  h->st.slots.clear ();
  h->has_prev = false; h->configured = false;       // reference: -cleanup drops the history (metaldeinterlacerenderer.m:422)
}

void vfhip_deinterlace_free (VfHipDeinterlace *h)
{
  if (!h) return;
  vfhip_deinterlace_cleanup (h);
  h->st.destroy ();
  delete h;
}

}  // extern "C"
