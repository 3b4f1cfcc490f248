// csrc/convertscale_kernels.h — device code of vfhipconvertscale (gfx950 / CDNA4 only).
//
// Replaces the reference's render pass convertScaleVertex + convertScaleFragment{RGBA,NV12,I420,UYVY,YUY2}
// (convertscale/metalconvertscale_shaders.h:48-198) and yuvToRGB (common/vfmetalshaders.m:40-79).
// Two arithmetic families (SURVEY.md finding 3):
//   gst-exact : the integer arithmetic of GStreamer 1.14 videoconvert+videoscale (bit-exact; oracle/gst114.c)
//   metal     : the float arithmetic of the reference shaders (convertscale_metal_kernels.h)
//
// Kernels in this file (streaming per-pixel kernels, no MFMA — there is no contraction; roofline = HBM):
//   k_cs_nv12_half : NV12 -> BGRA/RGBA at exactly 2:1 in both axes, bilinear (the BASELINE headline 2160p -> 1080p).
//                    Each lane owns 4 adjacent output pixels (one 16-byte store per row) and slides down a strip of
//                    `rows` output rows two rows per trip, keeping the horizontally up-sampled chroma row and the
//                    floor-average with its predecessor in registers, so every input byte is loaded once; the loads
//                    of the next row are issued before the current row is computed (register double buffer).
//                    Packed-byte ALU: v_lerp_u8 for (a+b+1)>>1 / (3a+b+2)>>2 and the w=128 vertical tap,
//                    v_mad_i32_i16 (op_sel) for the ORC mulhs matrix, v_sat_pk_u8_i16 for the clamps on planar
//                    [even, odd] byte pairs, v_dot4_u32_u8 for the horizontal tap.  DESIGN.md §5.1.
//   k_cs_taps      : NV12 / I420 -> BGRA/RGBA, bilinear at any ratio: one output pixel per lane, window loads +
//                    per-lane v_perm selectors, the same packed ORC pipeline.
//   k_cs_taps_strip: the same for launches large enough to fill the chip with a quarter of the waves: four output rows per lane, the
//                    chroma rows the two source rows share gathered and filtered once (its gathers, not its arithmetic, bound k_cs_taps).
//   k_cs_generic   : everything else that is gst-exact with an RGB output (RGB inputs, nearest, tiny frames):
//                    scalar, one output pixel per lane, 4 converted taps.
// 4:2:0 outputs: convertscale_planar_kernels.h; `metal` numerics: convertscale_metal_kernels.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vfhip {

struct CsParams {
  const uint8_t *in[3];
  int is[3];
  uint8_t *out;
  int os;
  size_t in_pitch, out_pitch;       // batch: frame k at base + k * pitch
  int in_w, in_h, out_w, out_h;     // full frame sizes
  int rx, ry, rw, rh;               // destination rectangle (== full frame unless add-borders)
  int in_fmt, out_rgba;             // out_rgba: 1 = RGBA byte order, 0 = BGRA
  int c[5];                         // ORC matrix p1..p5
  int cosited, nearest, vfirst, hscale_on;
  uint32_t hinc;                    // 16.16 horizontal increment (bilinear)
  const int *vtab;                  // bilinear: rh * {i0, i1, w, pad}; nearest: rh * {i, 0, 0, 0}
  const int *htab;                  // nearest: rw source columns
  uint32_t border;                  // border colour in output byte order
  int half_rows;                    // k_cs_nv12_half: output rows per lane (strip height)
  int xg_x, xg_y, xg_n, xg_chunk;   // XCD-aware 1-D launches (cs_xcd_block): blocks per row / column of a frame, blocks in all, blocks per XCD
  uint32_t xg_m_per, xg_l_per, xg_m_x, xg_l_x;   // n / (xg_x * xg_y) and n / xg_x as (umulhi (m, n) + n) >> l (cs_fastdiv, host)
};

// XCD-aware block order for kernels whose neighbouring blocks share source lines (k_cs_taps_strip: a block's 64 columns x 16 rows lean on 1.5 luma
// lines per row that its neighbours touch too).  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one and its L2), so a 1-D
// launch of 8 * chunk blocks gives block b the work item (b % 8) * chunk + b / 8 — each XCD ONE contiguous run (x fastest, then y, then frame) —
// and shared lines are fetched into one L2.  Returns false for the surplus blocks of the rounded-up grid.  Speed only: any mapping gives the same bytes.
__device__ __forceinline__ bool cs_xcd_block (const CsParams &p, int &bx, int &by, int &bz)
{
  const uint32_t bt = (blockIdx.x & 7u) * (uint32_t) p.xg_chunk + (blockIdx.x >> 3);
  if (bt >= (uint32_t) p.xg_n) return false;
  // two divisions by launch constants as multiply-high + shift (scalar: s_mul_hi_u32): the generic integer division is ~30 instructions, and a
  // block of this kernel is one strip of four rows per wave — the two divisions made every 2-tap down-scale 8-10 % slower when this went in
  const uint32_t z = (__umulhi (p.xg_m_per, bt) + bt) >> p.xg_l_per, r = bt - z * (uint32_t) (p.xg_x * p.xg_y);
  const uint32_t y = (__umulhi (p.xg_m_x, r) + r) >> p.xg_l_x;
  bz = (int) z; by = (int) y; bx = (int) (r - y * (uint32_t) p.xg_x);
  return true;
}

// ------------------------------------------------------------------------------------------------
// packed-byte helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lerp_u8 (uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_lerp (a, b, c); }
// (3a + b + 2) >> 2 per byte == (a + ((a + b) >> 1) + 1) >> 1   (exact)
__device__ __forceinline__ uint32_t filt31_u8 (uint32_t a, uint32_t b) { return lerp_u8 (a, lerp_u8 (a, b, 0u), 0x01010101u); }
__device__ __forceinline__ uint32_t avg_rnd_u8 (uint32_t a, uint32_t b) { return lerp_u8 (a, b, 0x01010101u); }
__device__ __forceinline__ uint32_t perm_b32 (uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm (hi, lo, sel); }
__device__ __forceinline__ uint32_t alignbyte (uint32_t hi, uint32_t lo, uint32_t n) { return __builtin_amdgcn_alignbyte (hi, lo, n); }

template <int HI>
__device__ __forceinline__ int mad_i32_i16 (uint32_t a, int coef, int acc)
{
  int d;
  // coef is wave-uniform (a kernel argument): "s" keeps it in an SGPR (one scalar operand per VOP3 is allowed on gfx9)
  if (HI) asm ("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "s"(coef), "v"(acc));
  else    asm ("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(coef), "v"(acc));
  return d;
}
__device__ __forceinline__ uint32_t sat_pk_u8_i16 (uint32_t v)
{
  uint32_t d;
  asm ("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(v));
  return d;
}

typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));
__device__ __forceinline__ u16x2 as_u16x2 (uint32_t v) { return __builtin_bit_cast (u16x2, v); }
__device__ __forceinline__ uint32_t as_u32 (u16x2 v) { return __builtin_bit_cast (uint32_t, v); }

// ------------------------------------------------------------------------------------------------
// k_cs_nv12_half
// ------------------------------------------------------------------------------------------------
struct CRow { uint32_t e01, e23, o01, o23; };   // horizontally up-sampled chroma of 8 source columns: even / odd columns

struct CRaw { uint2 v; uint32_t right, left; };      // raw chroma bytes of one row: 4 samples + right / left neighbour pair

// `plane` is wave-uniform, `off` = row * stride + 8 * cg as a 32-bit byte offset; `roff` / `loff` = byte distance of the
// right / left neighbour chroma pair, edge-replicated WITHOUT a branch (a divergent branch here forces vmcnt(0) and
// serialises the software pipeline): the last / first lane of a row re-reads its own outer pair (roff 6, loff 0).
template <bool COSITED>
__device__ __forceinline__ CRaw load_craw (const uint8_t *plane, uint32_t off, uint32_t roff, uint32_t loff)
{
  CRaw r;
  r.v = *reinterpret_cast<const uint2 *> (plane + off);                            // [U0V0U1V1][U2V2U3V3]
  r.right = *reinterpret_cast<const uint16_t *> (plane + (off + roff));
  r.left = 0;
  if (!COSITED) r.left = *reinterpret_cast<const uint16_t *> (plane + (off - loff));
  return r;
}

// horizontal chroma up-sampling of one row (GStreamer: horizontal first, then vertical)
template <bool COSITED>
__device__ __forceinline__ CRow hfilter (const CRaw &r)
{
  const uint2 v = r.v;
  const uint32_t r01 = alignbyte (v.y, v.x, 2);          // [U1V1U2V2]
  const uint32_t r23 = alignbyte (r.right, v.y, 2);      // [U3V3U4V4]
  CRow c;
  if (COSITED) {
    c.e01 = v.x; c.e23 = v.y;
    c.o01 = avg_rnd_u8 (v.x, r01); c.o23 = avg_rnd_u8 (v.y, r23);
  } else {
    const uint32_t l01 = (v.x << 16) | r.left;           // [U-1V-1U0V0]
    const uint32_t l23 = r01;                            // [U1V1U2V2]
    c.e01 = filt31_u8 (v.x, l01); c.e23 = filt31_u8 (v.y, l23);
    c.o01 = filt31_u8 (v.x, r01); c.o23 = filt31_u8 (v.y, r23);
  }
  return c;
}

// v_dot4_u32_u8 through the builtin, NOT inline asm: DOT results need 3 wait states before a different VALU
// reads them on gfx940/gfx950, and hipcc pads hazards only for instructions it models (an asm dot4 followed by
// v_perm returned stale bytes — found by the parity tests).
__device__ __forceinline__ uint32_t dot4_u8 (uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4 (a, b, c, false); }

// One output pixel's two source pixels of one source row: even column (chroma `uve`) and odd column (`uvo`);
// `ys` = [ye ye yo yo] byte-splatted luma (already ^0x80), uve/uvo = [U U V V] (already ^0x80).
// Produces planar byte pairs [X_even, X_odd] per channel (low 16 bits), saturated to u8.
__device__ __forceinline__ void orc_pair (uint32_t ys, uint32_t uve, uint32_t uvo, const int *c, int bias,
    uint32_t &bb, uint32_t &gg, uint32_t &rr)
{
  const int wye = mad_i32_i16<0> (ys, c[0], bias) & (int) 0xffff0000;
  const int wyo = mad_i32_i16<1> (ys, c[0], bias) & (int) 0xffff0000;
  const int tre = mad_i32_i16<1> (uve, c[1], wye), tro = mad_i32_i16<1> (uvo, c[1], wyo);
  const int tbe = mad_i32_i16<0> (uve, c[2], wye), tbo = mad_i32_i16<0> (uvo, c[2], wyo);
  const int tge = mad_i32_i16<1> (uve, c[4], mad_i32_i16<0> (uve, c[3], wye) & (int) 0xffff0000);
  const int tgo = mad_i32_i16<1> (uvo, c[4], mad_i32_i16<0> (uvo, c[3], wyo) & (int) 0xffff0000);
  // [even.hi16 | odd.hi16] -> saturate both int16 to u8 -> bytes [even, odd]
  bb = sat_pk_u8_i16 (perm_b32 ((uint32_t) tbo, (uint32_t) tbe, 0x07060302u));
  gg = sat_pk_u8_i16 (perm_b32 ((uint32_t) tgo, (uint32_t) tge, 0x07060302u));
  rr = sat_pk_u8_i16 (perm_b32 ((uint32_t) tro, (uint32_t) tre, 0x07060302u));
}

// the same with the two luma terms (mulhs (splat (y), c0) + bias, low half already cleared) handed in: k_cs_nv12_half takes them from a
// 256-entry table in LDS (half_row)
__device__ __forceinline__ void orc_pair_wy (int wye, int wyo, uint32_t uve, uint32_t uvo, const int *c,
    uint32_t &bb, uint32_t &gg, uint32_t &rr)
{
  const int tre = mad_i32_i16<1> (uve, c[1], wye), tro = mad_i32_i16<1> (uvo, c[1], wyo);
  const int tbe = mad_i32_i16<0> (uve, c[2], wye), tbo = mad_i32_i16<0> (uvo, c[2], wyo);
  const int tge = mad_i32_i16<1> (uve, c[4], mad_i32_i16<0> (uve, c[3], wye) & (int) 0xffff0000);
  const int tgo = mad_i32_i16<1> (uvo, c[4], mad_i32_i16<0> (uvo, c[3], wyo) & (int) 0xffff0000);
  bb = sat_pk_u8_i16 (perm_b32 ((uint32_t) tbo, (uint32_t) tbe, 0x07060302u));
  gg = sat_pk_u8_i16 (perm_b32 ((uint32_t) tgo, (uint32_t) tge, 0x07060302u));
  rr = sat_pk_u8_i16 (perm_b32 ((uint32_t) tro, (uint32_t) tre, 0x07060302u));
}

// per-lane constants of the fast path
struct HalfCtx {
  const uint8_t *yp, *uvp;
  uint8_t *op;
  uint32_t ys, cs, os, cx, roff, loff;
  int ch, yend;
  int c[5];
  uint32_t wgt[4];
  const uint32_t *wy;          // LDS: wy[b] = (mulhs (splat (b ^ 0x80), c0) + 128) << 16 for the RAW luma byte b
};

// 16 output bytes that nothing in the kernel reads again: a non-temporal store (measured on the headline config, three
// interleaved A/B rounds on one box: 235.0 -> 237.6 k frames/s; non-temporal LOADS on top changed nothing)
__device__ __forceinline__ void store_stream16 (uint8_t *dst, const uint32_t v[4])
{
  typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
  const v4u q = { v[0], v[1], v[2], v[3] };
  __builtin_nontemporal_store (q, reinterpret_cast<v4u *> (dst));
}

// One output row `y` of one lane (4 output pixels).  In: chroma state (hc = h-filtered chroma row y, mid_up = floor
// average of rows y-1 and y) and the prefetched raw rows of THIS row.  Out: the state for row y+1 and the prefetch of
// row y+1 (issued before this row's arithmetic: register double buffer).  Called twice per loop trip with the two
// register sets swapped, so the rotation costs no moves.
template <bool COSITED, bool RGBA, bool F0>
__device__ __forceinline__ void half_row (const HalfCtx &k, int y, const CRow &hc, const CRow &mid_up, const CRaw &craw, uint2 yt, uint2 yb,
    CRow &hn, CRow &mid_dn, CRaw &nraw, uint2 &nyt, uint2 &nyb)
{
  {
    const int yn = min (y + 1, k.yend - 1);              // the last row re-reads itself (never out of bounds)
    nraw = load_craw<COSITED> (k.uvp, __umul24 ((uint32_t) min (yn + 1, k.ch - 1), k.cs) + k.cx, k.roff, k.loff);
    const uint32_t yo = __umul24 ((uint32_t) (2 * yn), k.ys) + k.cx;
    nyt = *reinterpret_cast<const uint2 *> (k.yp + yo);
    nyb = *reinterpret_cast<const uint2 *> (k.yp + (yo + k.ys));
  }
  hn = hfilter<COSITED> (craw);
  mid_dn = { lerp_u8 (hc.e01, hn.e01, 0u), lerp_u8 (hc.e23, hn.e23, 0u), lerp_u8 (hc.o01, hn.o01, 0u), lerp_u8 (hc.o23, hn.o23, 0u) };
  const uint32_t K1 = 0x01010101u, X = 0x80808080u;
  // vertical chroma filter (3a+b+2)>>2: source row 2y leans on chroma row y-1, row 2y+1 on chroma row y+1
  const uint32_t te01 = lerp_u8 (hc.e01, mid_up.e01, K1) ^ X, te23 = lerp_u8 (hc.e23, mid_up.e23, K1) ^ X;
  const uint32_t to01 = lerp_u8 (hc.o01, mid_up.o01, K1) ^ X, to23 = lerp_u8 (hc.o23, mid_up.o23, K1) ^ X;
  const uint32_t be01 = lerp_u8 (hc.e01, mid_dn.e01, K1) ^ X, be23 = lerp_u8 (hc.e23, mid_dn.e23, K1) ^ X;
  const uint32_t bo01 = lerp_u8 (hc.o01, mid_dn.o01, K1) ^ X, bo23 = lerp_u8 (hc.o23, mid_dn.o23, K1) ^ X;
  // the luma term of each of the 16 source pixels from the LDS table: byte -> dword address with one shift and one mask (full rate),
  // the look-up on the LDS pipe — instead of xor + byte splat (v_perm) + v_mad_i32_i16 + mask per pixel on the VALU
  int wt[8], wb[8];
  {
    const uint32_t a0 = yt.x, a1 = yt.y, b0 = yb.x, b1 = yb.y;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t m = 0x3fcu;
      wt[j] = (int) k.wy[((j ? a0 >> (8 * j - 2) : a0 << 2) & m) >> 2]; wt[4 + j] = (int) k.wy[((j ? a1 >> (8 * j - 2) : a1 << 2) & m) >> 2];
      wb[j] = (int) k.wy[((j ? b0 >> (8 * j - 2) : b0 << 2) & m) >> 2]; wb[4 + j] = (int) k.wy[((j ? b1 >> (8 * j - 2) : b1 << 2) & m) >> 2];
    }
  }
  uint32_t out[4];
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t su = (n & 1) ? 0x03030202u : 0x01010000u;       // chroma pair n within its dword
    const uint32_t ten = n < 2 ? te01 : te23, ton = n < 2 ? to01 : to23, ben = n < 2 ? be01 : be23, bon = n < 2 ? bo01 : bo23;
    uint32_t bt, gt, rt, bbm, gbm, rbm;
    orc_pair_wy (wt[2 * n], wt[2 * n + 1], perm_b32 (0u, ten, su), perm_b32 (0u, ton, su), k.c, bt, gt, rt);
    orc_pair_wy (wb[2 * n], wb[2 * n + 1], perm_b32 (0u, ben, su), perm_b32 (0u, bon, su), k.c, bbm, gbm, rbm);
    // vertical 2-tap, w = 128: s1 + (((s2-s1)*128+128)>>8) == (s1+s2+1)>>1 on the [even, odd] byte pairs
    const uint32_t vb = avg_rnd_u8 (bt, bbm), vg = avg_rnd_u8 (gt, gbm), vr = avg_rnd_u8 (rt, rbm);
    // horizontal 2-tap (e*(256-f) + o*f) >> 8.  256 - f does not fit a byte when f == 0, so in general the weights are
    // (255 - f, f) and e is added through the accumulator; a wave none of whose lanes has an f == 0 (F0 false: 6 of 7.5 waves
    // of a 2160p -> 1080p row) multiplies by (256 - f, f) and saves the three masks per pixel
    const uint32_t hb = dot4_u8 (vb, k.wgt[n], F0 ? vb & 0xffu : 0u), hg = dot4_u8 (vg, k.wgt[n], F0 ? vg & 0xffu : 0u), hr = dot4_u8 (vr, k.wgt[n], F0 ? vr & 0xffu : 0u);
    const uint32_t lo = RGBA ? perm_b32 (hg, hr, 0x0c0c0501u) : perm_b32 (hg, hb, 0x0c0c0501u);      // [X>>8, G>>8, -, -]
    out[n] = perm_b32 (RGBA ? hb : hr, lo, 0x0d050100u);                                               // [X, G, Z, 0xff]
  }
  store_stream16 (k.op + (__umul24 ((uint32_t) y, k.os) + 2u * k.cx), out);
}

// one lane's strip: set-up, prologue loads and the row loop.  F0: some lane of the wave has a horizontal tap with f == 0
// (see half_row); the two instantiations are separate paths of the kernel so that each gets its own register allocation.
template <bool COSITED, bool RGBA, bool F0>
__device__ __forceinline__ void half_strip (const CsParams &p, int frame, int cg, int cgpr, int y0, int rows, const uint32_t *wy)
{
  HalfCtx k;
  k.wy = wy;
  k.yp = p.in[0] + (size_t) frame * p.in_pitch;               // wave-uniform plane bases; per-lane parts are 32-bit offsets
  k.uvp = p.in[1] + (size_t) frame * p.in_pitch;         // (global_load with SGPR base + VGPR offset)
  k.op = p.out + (size_t) frame * p.out_pitch;
  k.ys = (uint32_t) p.is[0]; k.cs = (uint32_t) p.is[1]; k.os = (uint32_t) p.os;
  k.cx = 8u * (uint32_t) cg;
  k.roff = cg == cgpr - 1 ? 6u : 8u; k.loff = cg == 0 ? 0u : 2u;
  k.ch = p.out_h;                                              // chroma rows == output rows at 2:1
  k.yend = min (y0 + rows, p.out_h);
#pragma unroll
  for (int i = 0; i < 5; i++) k.c[i] = p.c[i];
  // horizontal tap weights of this lane's 4 output pixels (row independent): bytes [255-f, f, 0, 0] (+ e through the accumulator),
  // or [256-f, f, 0, 0] when no lane of the wave has an f == 0
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t tt = (uint32_t) (cg * 4 + n) * p.hinc;
    const uint32_t f = (tt >> 8) & 0xffu;
    k.wgt[n] = ((F0 ? 255u : 256u) - f) | (f << 8);
  }
  CRow hcA, midA, hcB, midB;
  {
    const CRow hm = hfilter<COSITED> (load_craw<COSITED> (k.uvp, __umul24 ((uint32_t) max (y0 - 1, 0), k.cs) + k.cx, k.roff, k.loff));
    hcA = hfilter<COSITED> (load_craw<COSITED> (k.uvp, __umul24 ((uint32_t) y0, k.cs) + k.cx, k.roff, k.loff));
    // floor-average of chroma rows (j-1, j): the inner half of (3a+b+2)>>2; the (j, j+1) one is reused by the next row
    midA = { lerp_u8 (hcA.e01, hm.e01, 0u), lerp_u8 (hcA.e23, hm.e23, 0u), lerp_u8 (hcA.o01, hm.o01, 0u), lerp_u8 (hcA.o23, hm.o23, 0u) };
  }
  CRaw rawA = load_craw<COSITED> (k.uvp, __umul24 ((uint32_t) min (y0 + 1, k.ch - 1), k.cs) + k.cx, k.roff, k.loff), rawB;
  uint2 ytA = *reinterpret_cast<const uint2 *> (k.yp + (__umul24 ((uint32_t) (2 * y0), k.ys) + k.cx)), ytB;
  uint2 ybA = *reinterpret_cast<const uint2 *> (k.yp + (__umul24 ((uint32_t) (2 * y0 + 1), k.ys) + k.cx)), ybB;
  for (int y = y0; y < k.yend; y += 2) {
    half_row<COSITED, RGBA, F0> (k, y, hcA, midA, rawA, ytA, ybA, hcB, midB, rawB, ytB, ybB);
    if (y + 1 >= k.yend) break;                                // odd tail (only when out_h is odd)
    half_row<COSITED, RGBA, F0> (k, y + 1, hcB, midB, rawB, ytB, ybB, hcA, midA, rawA, ytA, ybA);
  }
}

// grid: 1-D, ceil(cgpr * strips / 256) blocks of 256 lanes per frame, frames back to back (cgpr = out_w / 4 column
// groups per row, strips of p.half_rows output rows).  (A "guided" variant that finished each batch with short strips
// to shorten the drain phase between back-to-back launches measured no gain — profiles/r01q_tail_ab.txt — and was dropped.)
template <bool COSITED, bool RGBA>
__global__ __launch_bounds__ (256, 8) void k_cs_nv12_half (const CsParams p)
{
  const int cgpr = p.out_w >> 2;
  const int b = blockIdx.x, rows = p.half_rows;
  const int strips = (p.out_h + rows - 1) / rows;
  const int bpf = (cgpr * strips + 255) >> 8;                  // blocks per frame (wave-uniform scalar arithmetic)
  const int frame = b / bpf;
  const int t = (b % bpf) * 256 + threadIdx.x;
  const uint32_t *wy = nullptr;
  __shared__ uint32_t wy_lut[256];
  {
    const uint32_t v = (uint32_t) threadIdx.x ^ 0x80u;               // the luma byte as ORC sees it; splat: both bytes of the 16-bit lane
    wy_lut[threadIdx.x] = (uint32_t) (mad_i32_i16<0> (v | (v << 8), p.c[0], 128 << 16) & (int) 0xffff0000);
  }
  __syncthreads ();
  wy = wy_lut;
  if (t >= cgpr * strips) return;
  const int strip = t / cgpr, cg = t - strip * cgpr;
  bool lane_f0 = false;
#pragma unroll
  for (int n = 0; n < 4; n++) lane_f0 |= ((((uint32_t) (cg * 4 + n) * p.hinc) >> 8) & 0xffu) == 0;
  if (__builtin_amdgcn_ballot_w64 (lane_f0) != 0) half_strip<COSITED, RGBA, true> (p, frame, cg, cgpr, strip * rows, rows, wy);      // wave-uniform
  else half_strip<COSITED, RGBA, false> (p, frame, cg, cgpr, strip * rows, rows, wy);
}

// ------------------------------------------------------------------------------------------------
// k_cs_i420_half<RGBA>: I420 -> BGRA / RGBA at exactly 2:1 in both axes, bilinear
// ------------------------------------------------------------------------------------------------
// GStreamer converts I420 with NEAREST-replicated chroma (its I420 fast path: oracle/gst114.c), so the four source pixels
// behind one output pixel share one (U, V): the three chroma terms of the ORC matrix are evaluated once per OUTPUT pixel,
// masked to their high halves, and added to each source pixel's luma term with full-rate v_add (the mulhs sums stay exact
// because every addend has a zero low half).  No chroma filter state crosses rows, so a lane simply walks its strip with
// the next row's loads in flight.  Same mapping, taps and packing as k_cs_nv12_half; ~25 % fewer VALU cycles per pixel.
template <bool RGBA>
__global__ __launch_bounds__ (256, 8) void k_cs_i420_half (const CsParams p)
{
  const int cgpr = p.out_w >> 2;
  const int b = blockIdx.x, rows = p.half_rows;
  const int strips = (p.out_h + rows - 1) / rows;
  const int bpf = (cgpr * strips + 255) >> 8;
  const int frame = b / bpf;
  const int t = (b % bpf) * 256 + threadIdx.x;
  if (t >= cgpr * strips) return;
  const int strip = t / cgpr, cg = t - strip * cgpr;
  const int y0 = strip * rows, yend = min (y0 + rows, p.out_h);
  const uint8_t *yp = p.in[0] + (size_t) frame * p.in_pitch, *up = p.in[1] + (size_t) frame * p.in_pitch, *vp = p.in[2] + (size_t) frame * p.in_pitch;
  uint8_t *op = p.out + (size_t) frame * p.out_pitch;
  const uint32_t ys = (uint32_t) p.is[0], us = (uint32_t) p.is[1], vs = (uint32_t) p.is[2], os = (uint32_t) p.os;
  const uint32_t cx = 8u * (uint32_t) cg, ccx = 4u * (uint32_t) cg;
  uint32_t wgt[4];
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const uint32_t tt = (uint32_t) (cg * 4 + n) * p.hinc;
    const uint32_t f = (tt >> 8) & 0xffu;
    wgt[n] = (255u - f) | (f << 8);
  }
  const uint32_t X = 0x80808080u;
  const int M = (int) 0xffff0000, bias = 128 << 16;
  auto load_row = [&] (int y, uint2 &yt, uint2 &yb, uint32_t &u4, uint32_t &v4) {
    const uint32_t yo = __umul24 ((uint32_t) (2 * y), ys) + cx;
    yt = *reinterpret_cast<const uint2 *> (yp + yo);
    yb = *reinterpret_cast<const uint2 *> (yp + (yo + ys));
    u4 = *reinterpret_cast<const uint32_t *> (up + (__umul24 ((uint32_t) y, us) + ccx));
    v4 = *reinterpret_cast<const uint32_t *> (vp + (__umul24 ((uint32_t) y, vs) + ccx));
  };
  uint2 yt, yb; uint32_t u4, v4;
  load_row (y0, yt, yb, u4, v4);
  for (int y = y0; y < yend; y++) {
    uint2 nyt, nyb; uint32_t nu4, nv4;
    load_row (min (y + 1, yend - 1), nyt, nyb, nu4, nv4);             // the last row re-reads itself (never out of bounds)
    const uint32_t ux = u4 ^ X, vx = v4 ^ X;
    const uint32_t yt0 = yt.x ^ X, yt1 = yt.y ^ X, yb0 = yb.x ^ X, yb1 = yb.y ^ X;
    uint32_t out[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
      // chroma terms of output pixel n: [U U V V] of chroma sample n, once for its 2x2 source pixels
      const uint32_t uv = perm_b32 (vx, ux, (uint32_t) n * 0x01010101u + 0x04040000u);
      const int cr = mad_i32_i16<1> (uv, p.c[1], 0) & M;                                   // mulhs (V, p2) << 16
      const int cb = mad_i32_i16<0> (uv, p.c[2], 0) & M;                                   // mulhs (U, p3) << 16
      const int cgn = mad_i32_i16<1> (uv, p.c[4], mad_i32_i16<0> (uv, p.c[3], 0) & M) & M; // (mulhs (U, p4) + mulhs (V, p5)) << 16
      const uint32_t sy = (n & 1) ? 0x03030202u : 0x01010000u;
      const uint32_t yst = perm_b32 (0u, n < 2 ? yt0 : yt1, sy), ysb = perm_b32 (0u, n < 2 ? yb0 : yb1, sy);
      // luma terms (+128 bias) of the four source pixels, low halves cleared
      const int wte = mad_i32_i16<0> (yst, p.c[0], bias) & M, wto = mad_i32_i16<1> (yst, p.c[0], bias) & M;
      const int wbe = mad_i32_i16<0> (ysb, p.c[0], bias) & M, wbo = mad_i32_i16<1> (ysb, p.c[0], bias) & M;
      // channel = luma + chroma (plain adds: exact, every addend has a zero low half), saturate the [even | odd] pair
#define VF_PAIR(we, wo, c) sat_pk_u8_i16 (perm_b32 ((uint32_t) ((wo) + (c)), (uint32_t) ((we) + (c)), 0x07060302u))
      const uint32_t bt = VF_PAIR (wte, wto, cb), gt = VF_PAIR (wte, wto, cgn), rt = VF_PAIR (wte, wto, cr);
      const uint32_t bbm = VF_PAIR (wbe, wbo, cb), gbm = VF_PAIR (wbe, wbo, cgn), rbm = VF_PAIR (wbe, wbo, cr);
#undef VF_PAIR
      const uint32_t vb = avg_rnd_u8 (bt, bbm), vg = avg_rnd_u8 (gt, gbm), vr = avg_rnd_u8 (rt, rbm);
      const uint32_t hb = dot4_u8 (vb, wgt[n], vb & 0xffu), hg = dot4_u8 (vg, wgt[n], vg & 0xffu), hr = dot4_u8 (vr, wgt[n], vr & 0xffu);
      const uint32_t lo = RGBA ? perm_b32 (hg, hr, 0x0c0c0501u) : perm_b32 (hg, hb, 0x0c0c0501u);
      out[n] = perm_b32 (RGBA ? hb : hr, lo, 0x0d050100u);
    }
    store_stream16 (op + (__umul24 ((uint32_t) y, os) + 2u * cx), out);
    yt = nyt; yb = nyb; u4 = nu4; v4 = nv4;
  }
}

// ------------------------------------------------------------------------------------------------
// k_cs_taps<I420>: NV12 / I420 -> BGRA / RGBA, 2-tap bilinear at ANY ratio (BASELINE configs[0], 1080p -> 720p, up-scales ...)
// ------------------------------------------------------------------------------------------------
// One output pixel per lane.  Its 2x2 taps are two ADJACENT source columns (xa, xa+1) on two source rows, so per
// source row everything the pixel needs is one 2-byte luma window and, per chroma row, one 8-byte window of four
// [U V] pairs (columns k-1 .. k+2): 6 loads per pixel instead of 36 byte gathers.  v_perm_b32 with a per-lane selector
// pulls the pairs out of the window; from there the arithmetic is the packed-byte ORC pipeline of the fast path
// (orc_pair on the [xa, xa+1] pair), then the two taps in either pass order with GStreamer's 8-bit weights.
typedef uint2 __attribute__ ((aligned (2))) uint2_a2;
typedef uint16_t __attribute__ ((aligned (1))) uint16_a1;
typedef uint32_t __attribute__ ((aligned (2))) uint32_a2;

template <bool I420>
__global__ __launch_bounds__ (256) void k_cs_taps (const CsParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = __builtin_amdgcn_readfirstlane ((int) (blockIdx.y * 4 + threadIdx.y));        // one row per wave: the row's taps are scalar loads
  if (x >= p.out_w || y >= p.out_h) return;
  const uint8_t *yp = p.in[0] + (size_t) blockIdx.z * p.in_pitch;
  const uint8_t *up = p.in[1] + (size_t) blockIdx.z * p.in_pitch;
  const uint8_t *vp = I420 ? p.in[2] + (size_t) blockIdx.z * p.in_pitch : nullptr;
  uint32_t *o = reinterpret_cast<uint32_t *> (p.out + (size_t) blockIdx.z * p.out_pitch + (size_t) y * p.os) + x;
  const int dx = x - p.rx, dy = y - p.ry;
  if (dx < 0 || dy < 0 || dx >= p.rw || dy >= p.rh) { *o = p.border; return; }
  // the row's taps are the same for the whole wave: as scalars, so that the row addresses below are scalar arithmetic
  const int i0 = __builtin_amdgcn_readfirstlane (p.vtab[4 * dy]), i1 = __builtin_amdgcn_readfirstlane (p.vtab[4 * dy + 1]), w = __builtin_amdgcn_readfirstlane (p.vtab[4 * dy + 2]);
  int xa = dx, xb = dx, f = 0;
  if (p.hscale_on) {
    const uint32_t t = (uint32_t) dx * p.hinc;
    xa = min ((int) (t >> 16), p.in_w - 1); f = (int) ((t >> 8) & 0xff); xb = min (xa + 1, p.in_w - 1);
  }
  const int cw = (p.in_w + 1) >> 1, chh = (p.in_h + 1) >> 1;
  const uint32_t X = 0x80808080u, K1 = 0x01010101u;
  const int bias = 128 << 16;

  // luma window: 2 bytes at ybase, per-lane selector -> [ya ya yb yb]
  const int ybase = min (xa, p.in_w - 2);
  const uint32_t oa = (uint32_t) (xa - ybase), ob = (uint32_t) (xb - ybase);
  const uint32_t sely = oa | (oa << 8) | (ob << 16) | (ob << 24);

  uint32_t sel_c = 0, sel_n = 0;
  int cbase = 0;
  const int ka = xa >> 1, kb = xb >> 1;
  if (!I420) {
    // chroma window: 8 bytes = pairs cbase .. cbase+3; selectors for [P(k_a) | P(k_b)] and their horizontal neighbours
    const int kna = (xa & 1) ? min (ka + 1, cw - 1) : (p.cosited ? ka : max (ka - 1, 0));
    const int knb = (xb & 1) ? min (kb + 1, cw - 1) : (p.cosited ? kb : max (kb - 1, 0));
    cbase = max (min (ka - 1, cw - 4), 0);
    const uint32_t fa = (uint32_t) (ka - cbase), fb = (uint32_t) (kb - cbase), ga = (uint32_t) (kna - cbase), gb = (uint32_t) (knb - cbase);
    // pair k of the window = bytes (2k, 2k + 1): selector halves 2k | (2k + 1) << 8, built with shifts (a 32-bit multiply is quarter rate)
    const uint32_t pc = (fa << 1) | (fb << 17), pn = (ga << 1) | (gb << 17);
    sel_c = (pc | (pc << 8)) + 0x01000100u;
    sel_n = (pn | (pn << 8)) + 0x01000100u;
  }

  uint32_t bb[2], gg[2], rr[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int r = t ? i1 : i0;
    const int j = r >> 1;
    uint32_t uv;                                        // [U(xa) V(xa) U(xb) V(xb)], already ^0x80
    if (I420) {                                         // GStreamer's I420 fast path: nearest-replicated chroma
      const uint32_t cu = (uint32_t) j * (uint32_t) p.is[1], cv = (uint32_t) j * (uint32_t) p.is[2];      // scalar row offsets + 32-bit lane offsets
      const uint32_t ua = up[cu + (uint32_t) ka], ub = up[cu + (uint32_t) kb];
      const uint32_t va = vp[cv + (uint32_t) ka], vb = vp[cv + (uint32_t) kb];
      uv = (ua | (va << 8) | (ub << 16) | (vb << 24)) ^ X;
    } else {
      const int jn = (r & 1) ? min (j + 1, chh - 1) : max (j - 1, 0);
      const uint2 w0 = *reinterpret_cast<const uint2_a2 *> (up + ((uint32_t) j * (uint32_t) p.is[1] + 2u * (uint32_t) cbase));      // scalar row offset + 32-bit lane offset
      const uint2 w1 = *reinterpret_cast<const uint2_a2 *> (up + ((uint32_t) jn * (uint32_t) p.is[1] + 2u * (uint32_t) cbase));
      const uint32_t a0 = perm_b32 (w0.y, w0.x, sel_c), n0 = perm_b32 (w0.y, w0.x, sel_n);
      const uint32_t a1 = perm_b32 (w1.y, w1.x, sel_c), n1 = perm_b32 (w1.y, w1.x, sel_n);
      // horizontal first: co-sited even columns have n == a, so (a+n+1)>>1 == a; then vertical (3a+b+2)>>2
      const uint32_t h0 = p.cosited ? lerp_u8 (a0, n0, K1) : filt31_u8 (a0, n0);
      const uint32_t h1 = p.cosited ? lerp_u8 (a1, n1, K1) : filt31_u8 (a1, n1);
      uv = filt31_u8 (h0, h1) ^ X;
    }
    const uint32_t yw = (uint32_t) *reinterpret_cast<const uint16_a1 *> (yp + ((uint32_t) r * (uint32_t) p.is[0] + (uint32_t) ybase)) ^ 0x8080u;
    orc_pair (perm_b32 (0u, yw, sely), perm_b32 (0u, uv, 0x01010000u), perm_b32 (0u, uv, 0x03030202u), p.c, bias, bb[t], gg[t], rr[t]);
  }

  // the two taps.  [a, b] byte pairs -> u16 lanes; vertical: (r0*(256-w) + r1*w + 128)>>8 ; horizontal: (a*(256-f) + b*f)>>8
  const uint32_t ww = (uint32_t) w | ((uint32_t) w << 16), wm = 0x01000100u - ww;
  const uint32_t fw = (uint32_t) (256 - f) | ((uint32_t) f << 16);
  uint32_t ch[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const uint32_t q0 = c == 0 ? bb[0] : (c == 1 ? gg[0] : rr[0]), q1 = c == 0 ? bb[1] : (c == 1 ? gg[1] : rr[1]);
    const u16x2 e0 = as_u16x2 (perm_b32 (0u, q0, 0x0c010c00u)), e1 = as_u16x2 (perm_b32 (0u, q1, 0x0c010c00u));   // [a | b]
    if (p.vfirst || !p.hscale_on) {
      u16x2 t = e0 * as_u16x2 (wm) + as_u16x2 (0x00800080u);
      t = e1 * as_u16x2 (ww) + t;
      const uint32_t v = (as_u32 (t) >> 8) & 0x00ff00ffu;                    // [va | vb]
      ch[c] = p.hscale_on ? (__builtin_amdgcn_udot2 (as_u16x2 (v), as_u16x2 (fw), 0u, false) >> 8) : (v & 0xffu);
    } else {
      const int h0 = (int) (__builtin_amdgcn_udot2 (e0, as_u16x2 (fw), 0u, false) >> 8);
      const int h1 = (int) (__builtin_amdgcn_udot2 (e1, as_u16x2 (fw), 0u, false) >> 8);
      ch[c] = (uint32_t) (h0 + (((h1 - h0) * w + 128) >> 8));
    }
  }
  *o = p.out_rgba ? (ch[2] | (ch[1] << 8) | (ch[0] << 16) | 0xff000000u) : (ch[0] | (ch[1] << 8) | (ch[2] << 16) | 0xff000000u);
}

// ------------------------------------------------------------------------------------------------
// k_cs_taps_strip<I420, COSITED, VFIRST, ROWS>: k_cs_taps with a lane walking ROWS output rows of its column
// ------------------------------------------------------------------------------------------------
// k_cs_taps at one pixel per lane is bound by the texture-address unit, not by arithmetic: a gather whose lanes are 3 source pixels
// apart costs 17 (2-byte) to 33 (8-byte) cycles per wave per CU against 6 for a coalesced load (tools/ubench/bperm_rate.hip), and an
// output row took two luma and FOUR chroma-window gathers — 167 cycles per CU where its arithmetic needs ~100.  Here
//   * the two source rows of a 2-tap output row are adjacent (i1 - i0 <= 1, host-checked), so their vertical chroma filters lean on
//     two or three distinct chroma rows, not four: row j0 and its neighbour always, a third only when i0 is even — and the horizontal
//     filter of a shared row is evaluated once (NV12: 2.5 instead of 4 window gathers on average, and 4-byte windows instead of 8-byte ones
//     when the chroma is co-sited; I420: the four byte gathers per source row become one 2-byte gather per plane, shared when both rows
//     sit on one chroma row);
//   * what depends on the output COLUMN only (tap position, weights, window selectors: a third of k_cs_taps' instructions) is computed
//     once per strip, and the next row's loads are issued before this row's arithmetic.
// Same arithmetic as k_cs_taps, bit for bit.  Contract (host): bilinear with horizontal scaling, no borders, in_w >= 8, adjacent vertical
// taps; the taps of a row are one 16-byte scalar load.
struct TapRow { uint2 a, b, c; uint32_t yw[2]; };      // NV12: chroma windows of rows j0, jn0 and the third row; I420: a = {U, V} pairs of row j0, b = of row j1

template <bool I420, bool COSITED, bool VFIRST, int ROWS>
__global__ __launch_bounds__ (256) void k_cs_taps_strip (const CsParams p)
{
  int bx, by, bz;
  if (!cs_xcd_block (p, bx, by, bz)) return;
  const int x = bx * 64 + threadIdx.x;
  const int strip = __builtin_amdgcn_readfirstlane ((int) (by * 4 + threadIdx.y));
  const int y0 = strip * ROWS, yend = min (y0 + ROWS, p.out_h);
  if (x >= p.out_w || y0 >= p.out_h) return;
  const uint8_t *yp = p.in[0] + (size_t) bz * p.in_pitch;
  const uint8_t *up = p.in[1] + (size_t) bz * p.in_pitch;
  const uint8_t *vp = I420 ? p.in[2] + (size_t) bz * p.in_pitch : nullptr;
  uint8_t *op = p.out + (size_t) bz * p.out_pitch + 4 * (size_t) x;
  const int4 *vt = reinterpret_cast<const int4 *> (p.vtab);
  const uint32_t tt = (uint32_t) x * p.hinc;
  const int xa = min ((int) (tt >> 16), p.in_w - 1), f = (int) ((tt >> 8) & 0xff), xb = min (xa + 1, p.in_w - 1);
  const int cw = (p.in_w + 1) >> 1, chh = (p.in_h + 1) >> 1;
  const uint32_t X = 0x80808080u, K1 = 0x01010101u;
  const int bias = 128 << 16;
  const int ybase = min (xa, p.in_w - 2);
  const uint32_t oa = (uint32_t) (xa - ybase), ob = (uint32_t) (xb - ybase);
  const uint32_t sely = oa | (oa << 8) | (ob << 16) | (ob << 24);
  uint32_t sel_c = 0, sel_n = 0;
  int cbase = 0;
  const int ka = xa >> 1, kb = xb >> 1;
  if (!I420) {
    const int kna = (xa & 1) ? min (ka + 1, cw - 1) : (COSITED ? ka : max (ka - 1, 0));
    const int knb = (xb & 1) ? min (kb + 1, cw - 1) : (COSITED ? kb : max (kb - 1, 0));
    // co-sited chroma: an even column takes its own pair, an odd one averages it with the next — the two taps (one even, one odd column)
    // touch pairs ka and ka + 1 only: a 4-byte window (half the gather cost of the 8-byte one the 3:1 filter of the other sitings needs)
    cbase = COSITED ? min (ka, cw - 2) : max (min (ka - 1, cw - 4), 0);
    const uint32_t fa = (uint32_t) (ka - cbase), fb = (uint32_t) (kb - cbase), ga = (uint32_t) (kna - cbase), gb = (uint32_t) (knb - cbase);
    const uint32_t pc = (fa << 1) | (fb << 17), pn = (ga << 1) | (gb << 17);
    sel_c = (pc | (pc << 8)) + 0x01000100u;
    sel_n = (pn | (pn << 8)) + 0x01000100u;
  } else {
    // one 2-byte window per plane at column cbase holds the samples of both taps: [U(ka) V(ka) U(kb) V(kb)] = perm ({V window, U window})
    cbase = min (ka, cw - 2);
    const uint32_t fa = (uint32_t) (ka - cbase), fb = (uint32_t) (kb - cbase);
    sel_c = fa | ((fa + 4u) << 8) | (fb << 16) | ((fb + 4u) << 24);
  }
  const uint32_t fw = (uint32_t) (256 - f) | ((uint32_t) f << 16);
  const uint32_t cs1 = (uint32_t) p.is[1], cs2 = I420 ? (uint32_t) p.is[2] : 0u;

  // chroma rows of a source row r: its own (j) and, for NV12's vertical filter, the one it leans on (jn)
  auto jn_of = [&] (int r) { const int j = r >> 1; return (r & 1) ? min (j + 1, chh - 1) : max (j - 1, 0); };
  auto win = [&] (int j) {
    const uint8_t *a = up + ((uint32_t) j * cs1 + 2u * (uint32_t) cbase);
    if (COSITED) return make_uint2 (*reinterpret_cast<const uint32_a2 *> (a), 0u);
    return (uint2) *reinterpret_cast<const uint2_a2 *> (a);
  };
  auto win420 = [&] (int j) {
    return make_uint2 ((uint32_t) *reinterpret_cast<const uint16_a1 *> (up + ((uint32_t) j * cs1 + (uint32_t) cbase)),
                       (uint32_t) *reinterpret_cast<const uint16_a1 *> (vp + ((uint32_t) j * cs2 + (uint32_t) cbase)));
  };
  auto load = [&] (int i0, int i1, TapRow &L) {
    const int j0 = i0 >> 1, j1 = i1 >> 1;
    L.b = make_uint2 (0u, 0u); L.c = make_uint2 (0u, 0u);
    if (I420) {
      L.a = win420 (j0);
      if (j1 != j0) L.b = win420 (j1);                                  // wave-uniform
    } else {
      const int jn0 = jn_of (i0), jn1 = jn_of (i1);
      const int jc = (j1 != j0 && j1 != jn0) ? j1 : jn1;               // the one row (at most) that source row i0 does not already use
      L.a = win (j0); L.b = win (jn0);
      if (jc != j0 && jc != jn0) L.c = win (jc);                        // wave-uniform
    }
    L.yw[0] = (uint32_t) *reinterpret_cast<const uint16_a1 *> (yp + ((uint32_t) i0 * (uint32_t) p.is[0] + (uint32_t) ybase));
    L.yw[1] = (uint32_t) *reinterpret_cast<const uint16_a1 *> (yp + ((uint32_t) i1 * (uint32_t) p.is[0] + (uint32_t) ybase));
  };
  auto hfilt = [&] (uint2 w) {                  // horizontal chroma filter of a window at this lane's two taps: co-sited even columns have n == a
    const uint32_t a = perm_b32 (w.y, w.x, sel_c), n = perm_b32 (w.y, w.x, sel_n);
    return COSITED ? lerp_u8 (a, n, K1) : filt31_u8 (a, n);
  };
  auto compute = [&] (const TapRow &L, int i0, int i1, int w, int y) {
    uint32_t uv[2];                             // [U(xa) V(xa) U(xb) V(xb)] of source rows i0 / i1, already ^0x80
    const int j0 = i0 >> 1, j1 = i1 >> 1;
    if (I420) {
      uv[0] = perm_b32 (L.a.y, L.a.x, sel_c) ^ X;
      uv[1] = j1 != j0 ? perm_b32 (L.b.y, L.b.x, sel_c) ^ X : uv[0];
    } else {
      const int jn0 = jn_of (i0), jn1 = jn_of (i1);
      const uint32_t ha = hfilt (L.a), hb = hfilt (L.b);
      uint32_t hc = 0;
      if ((j1 != j0 && j1 != jn0) || (jn1 != j0 && jn1 != jn0)) hc = hfilt (L.c);
      const uint32_t h1 = j1 == j0 ? ha : (j1 == jn0 ? hb : hc), hn1 = jn1 == j0 ? ha : (jn1 == jn0 ? hb : hc);
      uv[0] = filt31_u8 (ha, hb) ^ X;           // vertical (3a+b+2)>>2
      uv[1] = filt31_u8 (h1, hn1) ^ X;
    }
    uint32_t bb[2], gg[2], rr[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
      orc_pair (perm_b32 (0u, L.yw[t] ^ 0x8080u, sely), perm_b32 (0u, uv[t], 0x01010000u), perm_b32 (0u, uv[t], 0x03030202u), p.c, bias, bb[t], gg[t], rr[t]);
    const uint32_t ww = (uint32_t) w | ((uint32_t) w << 16), wm = 0x01000100u - ww;
    uint32_t ch[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint32_t q0 = c == 0 ? bb[0] : (c == 1 ? gg[0] : rr[0]), q1 = c == 0 ? bb[1] : (c == 1 ? gg[1] : rr[1]);
      const u16x2 e0 = as_u16x2 (perm_b32 (0u, q0, 0x0c010c00u)), e1 = as_u16x2 (perm_b32 (0u, q1, 0x0c010c00u));
      if (VFIRST) {
        u16x2 t = e0 * as_u16x2 (wm) + as_u16x2 (0x00800080u);
        t = e1 * as_u16x2 (ww) + t;
        const uint32_t v = (as_u32 (t) >> 8) & 0x00ff00ffu;
        ch[c] = __builtin_amdgcn_udot2 (as_u16x2 (v), as_u16x2 (fw), 0u, false) >> 8;
      } else {
        const int h0 = (int) (__builtin_amdgcn_udot2 (e0, as_u16x2 (fw), 0u, false) >> 8);
        const int h1 = (int) (__builtin_amdgcn_udot2 (e1, as_u16x2 (fw), 0u, false) >> 8);
        ch[c] = (uint32_t) (h0 + (((h1 - h0) * w + 128) >> 8));
      }
    }
    const uint32_t px = p.out_rgba ? (ch[2] | (ch[1] << 8) | (ch[0] << 16) | 0xff000000u) : (ch[0] | (ch[1] << 8) | (ch[2] << 16) | 0xff000000u);
    *reinterpret_cast<uint32_t *> (op + (uint32_t) y * (uint32_t) p.os) = px;
  };

  // every row's taps up front: scalar loads (a load after the first store could alias it and would become a per-lane load)
  int4 tp[ROWS + 1];
#pragma unroll
  for (int r = 0; r <= ROWS; r++) tp[r] = vt[min (y0 + r, p.out_h - 1)];
  TapRow L[2];
  load (tp[0].x, tp[0].y, L[0]);
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    if (y0 + r >= yend) break;                                       // wave-uniform
    load (tp[r + 1].x, tp[r + 1].y, L[(r + 1) & 1]);                 // the next row's loads are in flight during this row's arithmetic
    __builtin_amdgcn_sched_barrier (0);                              // (the scheduler otherwise sinks them below the first waits)
    compute (L[r & 1], tp[r].x, tp[r].y, tp[r].z, y0 + r);
  }
}

// ------------------------------------------------------------------------------------------------
// k_cs_generic: one output pixel per thread, 4 converted taps
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi (int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int splat16 (int x) { int v = ((x ^ 0x80) & 0xff) * 257; return v >= 32768 ? v - 65536 : v; }

// converted source pixel at (cx, cy) as 4 channel ints in OUTPUT byte order
__device__ __forceinline__ void cs_tap (const CsParams &p, const uint8_t *const in[3], int cx, int cy, int px[4])
{
  if (p.in_fmt == VFHIP_FORMAT_BGRA || p.in_fmt == VFHIP_FORMAT_RGBA) {
    const uint32_t v = *reinterpret_cast<const uint32_t *> (in[0] + (size_t) cy * p.is[0] + 4 * (size_t) cx);
    const bool swap = (p.in_fmt == VFHIP_FORMAT_RGBA) != (p.out_rgba != 0);
    px[0] = swap ? (v >> 16) & 0xff : v & 0xff;
    px[1] = (v >> 8) & 0xff;
    px[2] = swap ? v & 0xff : (v >> 16) & 0xff;
    px[3] = v >> 24;
    return;
  }
  const int cw = (p.in_w + 1) >> 1, chh = (p.in_h + 1) >> 1;
  const int j = cy >> 1, k = cx >> 1;
  int Y, U, V;
  if (p.in_fmt == VFHIP_FORMAT_UYVY || p.in_fmt == VFHIP_FORMAT_YUY2) {
    // packed 4:2:2 (UYVY: U Y0 V Y1, YUY2: Y0 U Y1 V): NV12's horizontal chroma rule, no vertical step (oracle/gst114.c
    // gst114_packed422_to_rgb)
    const int yuy2 = p.in_fmt == VFHIP_FORMAT_YUY2;
    const uint8_t *row = in[0] + (size_t) cy * p.is[0];
    const int kn = (cx & 1) ? min (k + 1, cw - 1) : max (k - 1, 0);
    const int u0 = row[4 * k + (yuy2 ? 1 : 0)], v0 = row[4 * k + (yuy2 ? 3 : 2)];
    const int un = row[4 * kn + (yuy2 ? 1 : 0)], vn = row[4 * kn + (yuy2 ? 3 : 2)];
    Y = row[2 * cx + (yuy2 ? 0 : 1)];
    if (p.cosited) { U = (cx & 1) ? (u0 + un + 1) >> 1 : u0; V = (cx & 1) ? (v0 + vn + 1) >> 1 : v0; }
    else { U = (3 * u0 + un + 2) >> 2; V = (3 * v0 + vn + 2) >> 2; }
  } else if (p.in_fmt == VFHIP_FORMAT_I420) {      // GStreamer's I420 fast path: nearest-replicated chroma
    Y = in[0][(size_t) cy * p.is[0] + cx];
    U = in[1][(size_t) j * p.is[1] + k];
    V = in[2][(size_t) j * p.is[2] + k];
  } else {
    Y = in[0][(size_t) cy * p.is[0] + cx];
    const int jn = (cy & 1) ? min (j + 1, chh - 1) : max (j - 1, 0);
    const uint8_t *r0 = in[1] + (size_t) j * p.is[1], *r1 = in[1] + (size_t) jn * p.is[1];
    const int kn = (cx & 1) ? min (k + 1, cw - 1) : max (k - 1, 0);
    int u0, v0, u1, v1;
    if (p.cosited) {
      if (cx & 1) { u0 = (r0[2 * k] + r0[2 * kn] + 1) >> 1; v0 = (r0[2 * k + 1] + r0[2 * kn + 1] + 1) >> 1;
                    u1 = (r1[2 * k] + r1[2 * kn] + 1) >> 1; v1 = (r1[2 * k + 1] + r1[2 * kn + 1] + 1) >> 1; }
      else { u0 = r0[2 * k]; v0 = r0[2 * k + 1]; u1 = r1[2 * k]; v1 = r1[2 * k + 1]; }
    } else {
      u0 = (3 * r0[2 * k] + r0[2 * kn] + 2) >> 2; v0 = (3 * r0[2 * k + 1] + r0[2 * kn + 1] + 2) >> 2;
      u1 = (3 * r1[2 * k] + r1[2 * kn] + 2) >> 2; v1 = (3 * r1[2 * k + 1] + r1[2 * kn + 1] + 2) >> 2;
    }
    U = (3 * u0 + u1 + 2) >> 2; V = (3 * v0 + v1 + 2) >> 2;
  }
  const int wy = (splat16 (Y) * p.c[0]) >> 16;
  const int r = clampi (wy + ((splat16 (V) * p.c[1]) >> 16), -128, 127) + 128;
  const int b = clampi (wy + ((splat16 (U) * p.c[2]) >> 16), -128, 127) + 128;
  const int g = clampi (wy + ((splat16 (U) * p.c[3]) >> 16) + ((splat16 (V) * p.c[4]) >> 16), -128, 127) + 128;
  px[0] = p.out_rgba ? r : b; px[1] = g; px[2] = p.out_rgba ? b : r; px[3] = 255;
}

__global__ __launch_bounds__ (256) void k_cs_generic (const CsParams p)
{
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= p.out_w || y >= p.out_h) return;
  const uint8_t *in[3] = { p.in[0] + (size_t) blockIdx.z * p.in_pitch,
                           p.in[1] ? p.in[1] + (size_t) blockIdx.z * p.in_pitch : nullptr,
                           p.in[2] ? p.in[2] + (size_t) blockIdx.z * p.in_pitch : nullptr };
  uint32_t *o = reinterpret_cast<uint32_t *> (p.out + (size_t) blockIdx.z * p.out_pitch + (size_t) y * p.os) + x;
  const int dx = x - p.rx, dy = y - p.ry;
  if (dx < 0 || dy < 0 || dx >= p.rw || dy >= p.rh) { *o = p.border; return; }
  int r[4];
  if (p.nearest) {
    cs_tap (p, in, p.htab[dx], p.vtab[4 * dy], r);
  } else {
    const int i0 = p.vtab[4 * dy], i1 = p.vtab[4 * dy + 1], w = p.vtab[4 * dy + 2];
    int xa = dx, xb = dx, f = 0;
    if (p.hscale_on) {
      const uint32_t t = (uint32_t) dx * p.hinc;
      xa = min ((int) (t >> 16), p.in_w - 1); f = (int) ((t >> 8) & 0xff); xb = min (xa + 1, p.in_w - 1);
    }
    int a0[4], b0[4], a1[4], b1[4];
    cs_tap (p, in, xa, i0, a0); cs_tap (p, in, xb, i0, b0);
    cs_tap (p, in, xa, i1, a1); cs_tap (p, in, xb, i1, b1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (p.vfirst) {
        const int va = a0[k] + (((a1[k] - a0[k]) * w + 128) >> 8);
        const int vb = b0[k] + (((b1[k] - b0[k]) * w + 128) >> 8);
        r[k] = p.hscale_on ? (va * (256 - f) + vb * f) >> 8 : va;
      } else {
        const int h0 = (a0[k] * (256 - f) + b0[k] * f) >> 8;
        const int h1 = (a1[k] * (256 - f) + b1[k] * f) >> 8;
        r[k] = h0 + (((h1 - h0) * w + 128) >> 8);
      }
    }
  }
  *o = (uint32_t) r[0] | ((uint32_t) r[1] << 8) | ((uint32_t) r[2] << 16) | ((uint32_t) r[3] << 24);
}

}  // namespace vfhip
