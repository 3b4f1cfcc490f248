// csrc/host_jpeg.hip — a JPEG decoder (sequential and progressive) for the overlay image loader.  Host code only and free of HIP headers (vfhip_host.h), so
// that it also builds as plain C++ under AddressSanitizer / UBSan for the corrupt-input tests (tests/test_parsers_asan.py).
//
// The reference hands overlay files to ImageIO (overlay/metaloverlayrenderer.m:166-245), which reads JPEG as well as PNG; logos and
// watermarks do arrive as JPEG.  Supported: baseline, extended-sequential and progressive Huffman JPEG (SOF0 / SOF1 / SOF2), 8-bit samples, greyscale or
// three components (YCbCr per JFIF, or RGB when an Adobe marker says so), sampling factors 1 and 2 in either direction (others
// replicate), interleaved or per-component scans, restart intervals.  Refused with VFHIP_ERR_UNSUPPORTED: arithmetic-coded, lossless,
// hierarchical and 12-bit streams, CMYK.  The arithmetic is the Independent JPEG Group's published decoder, stage for stage — the "islow"
// integer inverse DCT (Loeffler / Ligtenberg / Moschytz, 13-bit constants), triangle ("fancy") chroma up-sampling for h2v1 and h2v2,
// 16-bit fixed-point YCbCr -> RGB — so that the pixels equal libjpeg / libjpeg-turbo's (tests/test_jpeg_decode.py compares with Pillow);
// what CoreGraphics produces for the same file is not pinned.  Output: RGBA8, alpha 255, row 0 first.
#include "vfhip_host.h"
#include <cstdlib>
#include <new>

using namespace vfhip;

namespace vfhip {

namespace {

struct HuffTable {
  bool present = false;
  uint8_t bits[17] = { 0 };      // bits[l] = number of codes of length l
  uint8_t vals[256] = { 0 };
  int mincode[17], maxcode[18], valptr[17];
  bool build ()
  {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k; mincode[l] = code;
      code += bits[l]; k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      if (code > (1 << l)) return false;           // more codes of this length than the prefix property allows
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    return k <= 256;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;             // blocks per row / column of the coefficient array (padded to whole MCUs)
  int w = 0, hh = 0;              // true sample size: ceil (X * h / hmax), ceil (Y * v / vmax)
  std::vector<int16_t> coef;      // bw * bh blocks of 64, natural order
  std::vector<uint8_t> plane;     // (bw * 8) x (bh * 8) samples after the inverse DCT
  int pred = 0;
  bool latched = false;           // qn holds the table (set by the component's first scan)
  uint16_t qn[64] = { 0 };        // the component's quantisation table in natural order
};

struct BitReader {
  const uint8_t *p, *end;
  uint32_t acc = 0; int n = 0;
  bool hit_marker = false, truncated = false;        // truncated: the data ran out without a marker (a complete scan is followed by one)
  BitReader (const uint8_t *b, const uint8_t *e) : p (b), end (e) {}
  void fill ()
  {
    while (n <= 24) {
      int byte = 0;
      if (!hit_marker && p >= end) truncated = true;
      if (!hit_marker && p < end) {
        byte = *p;
        if (byte == 0xff) {
          if (p + 1 < end && p[1] == 0x00) p += 2;                      // stuffed zero
          else { hit_marker = true; byte = 0; }                          // a marker: feed zeros, the caller finds it
        } else p++;
      }
      acc |= (uint32_t) byte << (24 - n); n += 8;
    }
  }
  int bit () { if (n < 1) fill (); const int b = (int) (acc >> 31); acc <<= 1; n--; return b; }
  int bits (int k) { if (!k) return 0; if (n < k) fill (); const int v = (int) (acc >> (32 - k)); acc <<= k; n -= k; return v; }
  void align () { acc = 0; n = 0; }
};

static const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

static int decode_symbol (BitReader &br, const HuffTable &t)
{
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br.bit ();
    if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
  }
  return -1;
}
static int extend (int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// one block: DC difference + AC run / size pairs -> quantised coefficients in natural order (dequantised in the inverse DCT, in 32 bits)
static bool decode_block (BitReader &br, const HuffTable &dc, const HuffTable &ac, int &pred, int16_t *out)
{
  memset (out, 0, 64 * sizeof (int16_t));
  int s = decode_symbol (br, dc);
  if (s < 0 || s > 11) return false;
  pred += s ? extend (br.bits (s), s) : 0;
  if (pred < -32768 || pred > 32767) return false;
  out[0] = (int16_t) pred;
  for (int k = 1; k < 64;) {
    const int rs = decode_symbol (br, ac);
    if (rs < 0) return false;
    const int r = rs >> 4, sz = rs & 15;
    if (!sz) { if (r == 15) { k += 16; continue; } break; }              // ZRL / end of block
    k += r;
    if (k > 63) return false;
    out[kZigzag[k]] = (int16_t) extend (br.bits (sz), sz);
    k++;
  }
  return true;
}

// ---- progressive scans (SOF2): every scan adds to the coefficients that earlier scans left in the component's array ---------------
// DC scans (spectral band 0..0): the first sends the difference, shifted down by Al; each refinement one more bit per block.
static bool prog_dc (BitReader &br, const HuffTable &dc, int &pred, int16_t *blk, int ah, int al)
{
  if (ah == 0) {
    const int s = decode_symbol (br, dc);
    if (s < 0 || s > 11) return false;
    pred += s ? extend (br.bits (s), s) : 0;
    if (pred < -32768 || pred > 32767) return false;
    blk[0] = (int16_t) (pred * (1 << al));
  } else if (br.bit ()) blk[0] = (int16_t) (blk[0] | (1 << al));
  return true;
}
// AC scans (band ss..se of ONE component): first pass — run / size pairs with end-of-band runs that span blocks
static bool prog_ac_first (BitReader &br, const HuffTable &ac, int16_t *blk, int ss, int se, int al, int &eobrun)
{
  if (eobrun > 0) { eobrun--; return true; }
  for (int k = ss; k <= se;) {
    const int rs = decode_symbol (br, ac);
    if (rs < 0) return false;
    const int r = rs >> 4, sz = rs & 15;
    if (sz) {
      k += r;
      if (k > se) return false;
      blk[kZigzag[k]] = (int16_t) (extend (br.bits (sz), sz) * (1 << al));
      k++;
    } else if (r == 15) k += 16;
    else { eobrun = (1 << r) - 1 + (r ? br.bits (r) : 0); break; }
  }
  return true;
}
// ... refinement: one correction bit for every coefficient that is already non-zero, newly non-zero ones (+-1 << al) in between
static bool prog_ac_refine (BitReader &br, const HuffTable &ac, int16_t *blk, int ss, int se, int al, int &eobrun)
{
  const int p1 = 1 << al, m1 = -(1 << al);
  int k = ss;
  auto correct = [&] (int16_t &c) { if (br.bit () && !(c & p1)) c = (int16_t) (c + (c >= 0 ? p1 : m1)); };
  if (eobrun == 0) {
    for (; k <= se; k++) {
      const int rs = decode_symbol (br, ac);
      if (rs < 0) return false;
      int r = rs >> 4;
      const int sz = rs & 15;
      int value = 0;
      if (sz) {
        if (sz != 1) return false;
        value = br.bit () ? p1 : m1;
      } else if (r != 15) { eobrun = (1 << r) + (r ? br.bits (r) : 0); break; }
      // skip r still-zero coefficients, correcting the non-zero ones on the way
      for (; k <= se; k++) {
        int16_t &c = blk[kZigzag[k]];
        if (c) correct (c);
        else if (--r < 0) break;
      }
      if (value) { if (k > se) return false; blk[kZigzag[k]] = (int16_t) value; }
    }
  }
  if (eobrun > 0) {
    for (; k <= se; k++) { int16_t &c = blk[kZigzag[k]]; if (c) correct (c); }
    eobrun--;
  }
  return true;
}

// the IJG "islow" inverse DCT on one block (coefficients times the quantisation table `q`, both in natural order) -> 8 x 8 samples
static void idct_islow (const int16_t *cin, const uint16_t *q, uint8_t *out, size_t stride)
{
  typedef long long I;                   // 64-bit intermediates: the values of a valid stream fit 32 bits (the IJG code uses them), corrupt data must not overflow
  I in[64];
  for (int k = 0; k < 64; k++) in[k] = (I) cin[k] * (I) q[k];
  const int CB = 13, P1 = 2;
  const I F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819,
            F2562 = 20995, F3072 = 25172;
  I ws[64];
  for (int c = 0; c < 8; c++) {
    const I *i = in + c;
    I *w = ws + c;
    if (!(i[8] | i[16] | i[24] | i[32] | i[40] | i[48] | i[56])) {
      const I dc = i[0] * (1 << P1);
      for (int r = 0; r < 8; r++) w[8 * r] = dc;
      continue;
    }
    I z2 = i[16], z3 = i[48];
    I z1 = (z2 + z3) * F0541;
    I tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = i[0]; z3 = i[32];
    I tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
    const I tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = i[56]; tmp1 = i[40]; tmp2 = i[24]; tmp3 = i[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    I z4 = tmp1 + tmp3;
    const I z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int sh = CB - P1; const I rnd = (I) 1 << (sh - 1);
    w[0] = (tmp10 + tmp3 + rnd) >> sh; w[56] = (tmp10 - tmp3 + rnd) >> sh;
    w[8] = (tmp11 + tmp2 + rnd) >> sh; w[48] = (tmp11 - tmp2 + rnd) >> sh;
    w[16] = (tmp12 + tmp1 + rnd) >> sh; w[40] = (tmp12 - tmp1 + rnd) >> sh;
    w[24] = (tmp13 + tmp0 + rnd) >> sh; w[32] = (tmp13 - tmp0 + rnd) >> sh;
  }
  for (int r = 0; r < 8; r++) {
    const I *w = ws + 8 * r;
    uint8_t *o = out + stride * (size_t) r;
    I z2 = w[2], z3 = w[6];
    I z1 = (z2 + z3) * F0541;
    I tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    I tmp0 = (w[0] + w[4]) * (1 << CB), tmp1 = (w[0] - w[4]) * (1 << CB);
    const I tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    I z4 = tmp1 + tmp3;
    const I z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int sh = CB + P1 + 3; const I rnd = (I) 1 << (sh - 1);
    auto lim = [] (I v) { v += 128; return (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v)); };
    o[0] = lim ((tmp10 + tmp3 + rnd) >> sh); o[7] = lim ((tmp10 - tmp3 + rnd) >> sh);
    o[1] = lim ((tmp11 + tmp2 + rnd) >> sh); o[6] = lim ((tmp11 - tmp2 + rnd) >> sh);
    o[2] = lim ((tmp12 + tmp1 + rnd) >> sh); o[5] = lim ((tmp12 - tmp1 + rnd) >> sh);
    o[3] = lim ((tmp13 + tmp0 + rnd) >> sh); o[4] = lim ((tmp13 - tmp0 + rnd) >> sh);
  }
}

static uint16_t be16 (const uint8_t *p) { return (uint16_t) ((p[0] << 8) | p[1]); }

// component plane (true size cw x ch, row stride cs) -> full size W x H
static void upsample (const Component &c, int hmax, int vmax, int W, int H, std::vector<uint8_t> &full)
{
  full.assign ((size_t) W * H, 0);
  const size_t cs = (size_t) c.bw * 8;
  const int cw = c.w, ch = c.hh;
  const uint8_t *src = c.plane.data ();
  const int fv = vmax / c.v;
  // (the frame header refuses fractional ratios; the 1:1 copy reads W x H samples and so must never be taken for a component plane that is smaller)
  if (c.h == hmax && c.v == vmax) {
    for (int y = 0; y < H; y++) memcpy (&full[(size_t) y * W], src + cs * (size_t) y, (size_t) W);
    return;
  }
  if (hmax == 2 * c.h && (vmax == c.v || vmax == 2 * c.v) && cw > 2) {       // (libjpeg filters only components wider than two samples)
    // triangle filter: each output sample is 3/4 of the nearer and 1/4 of the farther input sample (h2v2: in both directions, 16ths)
    std::vector<int> sum ((size_t) cw);
    for (int y = 0; y < H; y++) {
      const int iy = fv == 2 ? y >> 1 : y;
      const uint8_t *r0 = src + cs * (size_t) (iy < ch ? iy : ch - 1);
      uint8_t *o = &full[(size_t) y * W];
      if (fv == 1) {
        for (int x = 0; x < W; x++) {
          const int i = x >> 1;
          const int a = r0[i < cw ? i : cw - 1];
          if (!(x & 1)) o[x] = i == 0 ? (uint8_t) a : (uint8_t) ((a * 3 + r0[i - 1] + 1) >> 2);
          else o[x] = i >= cw - 1 ? (uint8_t) a : (uint8_t) ((a * 3 + r0[i + 1] + 2) >> 2);
        }
      } else {
        int ny = (y & 1) ? iy + 1 : iy - 1;                      // the farther row: above for even output rows, below for odd ones
        ny = ny < 0 ? 0 : (ny > ch - 1 ? ch - 1 : ny);
        const uint8_t *r1 = src + cs * (size_t) ny;
        for (int i = 0; i < cw; i++) sum[(size_t) i] = r0[i] * 3 + r1[i];
        for (int x = 0; x < W; x++) {
          int i = x >> 1;
          if (i > cw - 1) i = cw - 1;
          const int t = sum[(size_t) i];
          if (!(x & 1)) o[x] = i == 0 ? (uint8_t) ((t * 4 + 8) >> 4) : (uint8_t) ((t * 3 + sum[(size_t) i - 1] + 8) >> 4);
          else o[x] = i >= cw - 1 ? (uint8_t) ((t * 4 + 7) >> 4) : (uint8_t) ((t * 3 + sum[(size_t) i + 1] + 7) >> 4);
        }
      }
    }
    return;
  }
  for (int y = 0; y < H; y++) {                                   // any other ratio: replication
    const int iy = (int) ((long long) y * c.v / vmax);
    const uint8_t *r = src + cs * (size_t) (iy < ch ? iy : ch - 1);
    for (int x = 0; x < W; x++) { const int ix = (int) ((long long) x * c.h / hmax); full[(size_t) y * W + x] = r[ix < cw ? ix : cw - 1]; }
  }
}

static int decode_jpeg_impl (const char *path, std::vector<uint8_t> &rgba, int *width, int *height)
{
  FILE *f = fopen (path, "rb");
  if (!f) return set_error (VFHIP_ERR_INVALID, "cannot open %s", path);
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread (buf, 1, sizeof buf, f)) > 0) file.insert (file.end (), buf, buf + n);
  fclose (f);
  if (file.size () < 4 || file[0] != 0xff || file[1] != 0xd8) return set_error (VFHIP_ERR_UNSUPPORTED, "%s is not a JPEG file", path);
  uint16_t qt[4][64] = { { 0 } };
  bool have_qt[4] = { false, false, false, false };
  HuffTable hdc[4], hac[4];
  std::vector<Component> comp;
  int W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, adobe_transform = -1, n_scans = 0;
  const int kMaxScans = 128;
  bool have_sof = false, scans = false, eoi = false, progressive = false;
  size_t pos = 2;
  const uint8_t *data = file.data ();
  const size_t size = file.size ();
  while (!eoi) {
    while (pos < size && data[pos] != 0xff) pos++;               // (garbage between segments is skipped like libjpeg does)
    while (pos < size && data[pos] == 0xff) pos++;
    if (pos >= size) break;
    const int m = data[pos++];
    if (m == 0xd9) { eoi = true; break; }
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;         // standalone markers
    if (pos + 2 > size) return set_error (VFHIP_ERR_INVALID, "%s: truncated JPEG segment", path);
    const size_t len = be16 (data + pos);
    if (len < 2 || pos + len > size) return set_error (VFHIP_ERR_INVALID, "%s: truncated JPEG segment", path);
    const uint8_t *s = data + pos + 2;
    const size_t sl = len - 2;
    pos += len;
    if (m == 0xdb) {                                              // DQT
      for (size_t i = 0; i < sl;) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        i++;
        if (tq > 3 || pq > 1 || i + (size_t) 64 * (pq + 1) > sl) return set_error (VFHIP_ERR_INVALID, "%s: bad quantisation table", path);
        for (int k = 0; k < 64; k++) { qt[tq][k] = pq ? be16 (s + i + 2 * k) : s[i + k]; }
        have_qt[tq] = true;
        i += (size_t) 64 * (pq + 1);
      }
    } else if (m == 0xc4) {                                       // DHT
      for (size_t i = 0; i < sl;) {
        if (i + 17 > sl) return set_error (VFHIP_ERR_INVALID, "%s: bad Huffman table", path);
        const int tc = s[i] >> 4, th = s[i] & 15;
        if (tc > 1 || th > 3) return set_error (VFHIP_ERR_INVALID, "%s: bad Huffman table id", path);
        HuffTable &t = tc ? hac[th] : hdc[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) { t.bits[l] = s[i + l]; total += t.bits[l]; }
        i += 17;
        if (total > 256 || i + (size_t) total > sl) return set_error (VFHIP_ERR_INVALID, "%s: bad Huffman table", path);
        memset (t.vals, 0, sizeof t.vals);
        memcpy (t.vals, s + i, (size_t) total);
        i += (size_t) total;
        if (!t.build ()) return set_error (VFHIP_ERR_INVALID, "%s: inconsistent Huffman table", path);
        t.present = true;
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {             // SOF0 / SOF1 / SOF2 (progressive)
      if (have_sof) return set_error (VFHIP_ERR_INVALID, "%s: two frame headers", path);
      if (sl < 6) return set_error (VFHIP_ERR_INVALID, "%s: bad frame header", path);
      if (s[0] != 8) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: %d-bit JPEG samples are not supported", path, s[0]);
      H = be16 (s + 1); W = be16 (s + 3);
      const int nf = s[5];
      if (W <= 0 || H <= 0 || W > 16384 || H > 16384 || (size_t) W * (size_t) H > ((size_t) 64 << 20))
        return set_error (VFHIP_ERR_INVALID, "%s: bad JPEG size %dx%d (at most 16384 per side, 64 Mpixel)", path, W, H);
      if (nf == 4) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: four-component (CMYK) JPEG files are not supported", path);
      if ((nf != 1 && nf != 3) || sl < (size_t) 6 + 3 * (size_t) nf) return set_error (VFHIP_ERR_INVALID, "%s: bad frame header", path);
      comp.resize ((size_t) nf);
      for (int k = 0; k < nf; k++) {
        Component &c = comp[(size_t) k];
        c.id = s[6 + 3 * k]; c.h = s[7 + 3 * k] >> 4; c.v = s[7 + 3 * k] & 15; c.tq = s[8 + 3 * k];
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return set_error (VFHIP_ERR_INVALID, "%s: bad component in the frame header", path);
        hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
      }
      // fractional ratios (3:2, 4:3): libjpeg has no up-sampler for them either (JERR_FRACT_SAMPLE_NOTIMPL)
      for (const Component &c : comp)
        if (hmax % c.h || vmax % c.v)
          return set_error (VFHIP_ERR_UNSUPPORTED, "%s: fractional sampling ratio %d:%d x %d:%d is not supported", path, hmax, c.h, vmax, c.v);
      const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      for (Component &c : comp) {
        c.bw = mcux * c.h; c.bh = mcuy * c.v;
        c.w = (W * c.h + hmax - 1) / hmax; c.hh = (H * c.v + vmax - 1) / vmax;
        c.coef.assign ((size_t) c.bw * c.bh * 64, 0);
      }
      have_sof = true; progressive = m == 0xc2;
    } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
      return set_error (VFHIP_ERR_UNSUPPORTED, "%s: only Huffman-coded sequential and progressive JPEG is supported (frame type 0x%02x: lossless, hierarchical or arithmetic-coded)", path, m);
    } else if (m == 0xcc) {
      return set_error (VFHIP_ERR_UNSUPPORTED, "%s: arithmetic-coded JPEG is not supported", path);
    } else if (m == 0xdd) {                                       // DRI
      if (sl < 2) return set_error (VFHIP_ERR_INVALID, "%s: bad restart interval", path);
      restart = be16 (s);
    } else if (m == 0xee) {                                       // APP14 "Adobe": colour transform flag
      if (sl >= 12 && !memcmp (s, "Adobe", 5)) adobe_transform = s[11];
    } else if (m == 0xda) {                                       // SOS + entropy-coded data
      if (!have_sof) return set_error (VFHIP_ERR_INVALID, "%s: scan before the frame header", path);
      // every scan walks all blocks of its components whatever its size in the file: bound the scans (libjpeg-turbo's scan limit, same reason) and
      // refuse one that carries no entropy-coded byte at all
      if (++n_scans > kMaxScans) return set_error (VFHIP_ERR_UNSUPPORTED, "%s: more than %d scans", path, kMaxScans);
      if (pos >= size || (data[pos] == 0xff && (pos + 1 >= size || data[pos + 1] != 0x00)))
        return set_error (VFHIP_ERR_INVALID, "%s: scan without entropy-coded data", path);
      if (sl < 1) return set_error (VFHIP_ERR_INVALID, "%s: bad scan header", path);
      const int ns = s[0];
      if (ns < 1 || ns > (int) comp.size () || sl < (size_t) 1 + 2 * (size_t) ns + 3) return set_error (VFHIP_ERR_INVALID, "%s: bad scan header", path);
      Component *sc[3];
      for (int k = 0; k < ns; k++) {
        sc[k] = nullptr;
        for (Component &c : comp) if (c.id == s[1 + 2 * k]) sc[k] = &c;
        if (!sc[k]) return set_error (VFHIP_ERR_INVALID, "%s: scan names an unknown component", path);
        sc[k]->td = s[2 + 2 * k] >> 4; sc[k]->ta = s[2 + 2 * k] & 15;
        sc[k]->pred = 0;
      }
      const int ss = s[1 + 2 * ns], se = s[2 + 2 * ns], ah = s[3 + 2 * ns] >> 4, al = s[3 + 2 * ns] & 15;
      if (!progressive) { if (ss != 0 || se != 63 || ah || al) return set_error (VFHIP_ERR_INVALID, "%s: spectral selection in a sequential JPEG", path); }
      else if (ss > se || se > 63 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || (ah && ah != al + 1))
        return set_error (VFHIP_ERR_INVALID, "%s: bad progressive scan parameters", path);
      for (int k = 0; k < ns; k++) {
        const bool need_dc = ss == 0 && ah == 0, need_ac = progressive ? ss > 0 : true;
        if (sc[k]->td > 3 || sc[k]->ta > 3 || (need_dc && !hdc[sc[k]->td].present) || (need_ac && !hac[sc[k]->ta].present) || !have_qt[sc[k]->tq])
          return set_error (VFHIP_ERR_INVALID, "%s: scan uses a table the file does not define", path);
        if (!sc[k]->latched) for (int z = 0; z < 64; z++) sc[k]->qn[kZigzag[z]] = qt[sc[k]->tq][z];    // (the table in force when the component's FIRST scan starts)
        sc[k]->latched = true;
      }
      int eobrun = 0;
      // the entropy-coded segment runs to the next marker that is not a restart marker
      BitReader br (data + pos, data + size);
      int mx, my;
      if (ns == 1) { mx = (sc[0]->w + 7) / 8; my = (sc[0]->hh + 7) / 8; }        // a single-component scan covers the component's own blocks
      else { mx = (W + 8 * hmax - 1) / (8 * hmax); my = (H + 8 * vmax - 1) / (8 * vmax); }
      int todo = restart, rst = 0;
      for (int y = 0; y < my; y++)
        for (int x = 0; x < mx; x++) {
          if (restart && todo == 0) {
            // restart marker: byte-align, expect RSTn, reset the predictors
            br.align ();
            const uint8_t *q = br.p;
            while (q < br.end && *q != 0xff) q++;
            while (q < br.end && *q == 0xff) q++;
            if (q >= br.end || *q != 0xd0 + rst) return set_error (VFHIP_ERR_INVALID, "%s: missing restart marker", path);
            br.p = q + 1; br.hit_marker = false;
            rst = (rst + 1) & 7; todo = restart;
            for (int k = 0; k < ns; k++) sc[k]->pred = 0;
            eobrun = 0;
          }
          for (int k = 0; k < ns; k++) {
            Component &c = *sc[k];
            const int nh = ns == 1 ? 1 : c.h, nv = ns == 1 ? 1 : c.v;
            for (int by = 0; by < nv; by++)
              for (int bx = 0; bx < nh; bx++) {
                const int gx = x * nh + bx, gy = y * nv + by;
                int16_t tmp[64];
                bool okb;
                if (!progressive) {
                  okb = decode_block (br, hdc[c.td], hac[c.ta], c.pred, tmp);
                  if (okb && gx < c.bw && gy < c.bh) memcpy (&c.coef[((size_t) gy * c.bw + gx) * 64], tmp, sizeof tmp);
                } else {
                  int16_t *blk = (gx < c.bw && gy < c.bh) ? &c.coef[((size_t) gy * c.bw + gx) * 64] : (memset (tmp, 0, sizeof tmp), tmp);
                  if (ss == 0) okb = prog_dc (br, hdc[c.td], c.pred, blk, ah, al);
                  else okb = ah ? prog_ac_refine (br, hac[c.ta], blk, ss, se, al, eobrun) : prog_ac_first (br, hac[c.ta], blk, ss, se, al, eobrun);
                }
                if (!okb) return set_error (VFHIP_ERR_INVALID, "%s: corrupt JPEG data", path);
              }
          }
          if (restart) todo--;
        }
      // continue after the entropy-coded data
      pos = (size_t) (br.p - data);
      if (pos > size) pos = size;
      if (br.truncated) return set_error (VFHIP_ERR_INVALID, "%s: truncated JPEG data", path);
      // the next marker that is not a restart marker (the bit reader stops in front of a marker; padding bits may remain before it)
      while (pos >= 1 && pos < size && !(data[pos] == 0xff && pos + 1 < size && data[pos + 1] != 0x00 && !(data[pos + 1] >= 0xd0 && data[pos + 1] <= 0xd7))) pos++;
      scans = true;
    }
    // every other segment (APPn, COM, ...) is skipped
  }
  if (!have_sof || !scans) return set_error (VFHIP_ERR_INVALID, "%s: JPEG file without image data", path);
  for (Component &c : comp) {
    c.plane.assign ((size_t) c.bw * 8 * (size_t) c.bh * 8, 0);
    const size_t stride = (size_t) c.bw * 8;
    for (int by = 0; by < c.bh; by++)
      for (int bx = 0; bx < c.bw; bx++) idct_islow (&c.coef[((size_t) by * c.bw + bx) * 64], c.qn, &c.plane[(size_t) by * 8 * stride + (size_t) bx * 8], stride);
    std::vector<int16_t> ().swap (c.coef);
  }
  rgba.assign ((size_t) W * H * 4, 255);
  if (comp.size () == 1) {
    const size_t stride = (size_t) comp[0].bw * 8;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) { const uint8_t v = comp[0].plane[(size_t) y * stride + x]; uint8_t *d = &rgba[((size_t) y * W + x) * 4]; d[0] = d[1] = d[2] = v; }
  } else {
    std::vector<uint8_t> p0, p1, p2;
    upsample (comp[0], hmax, vmax, W, H, p0); upsample (comp[1], hmax, vmax, W, H, p1); upsample (comp[2], hmax, vmax, W, H, p2);
    const bool rgb = adobe_transform == 0 || (adobe_transform < 0 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
    auto lim = [] (int v) { return (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (size_t i = 0; i < (size_t) W * H; i++) {
      uint8_t *d = &rgba[i * 4];
      if (rgb) { d[0] = p0[i]; d[1] = p1[i]; d[2] = p2[i]; continue; }
      // JFIF YCbCr -> RGB in 16-bit fixed point (FIX (1.40200) = 91881, FIX (1.77200) = 116130, FIX (0.71414) = 46802, FIX (0.34414) = 22554)
      const int y = p0[i], cb = p1[i] - 128, cr = p2[i] - 128;
      d[0] = lim (y + ((91881 * cr + 32768) >> 16));
      d[1] = lim (y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
      d[2] = lim (y + ((116130 * cb + 32768) >> 16));
    }
  }
  *width = W; *height = H;
  return VFHIP_OK;
}

}  // namespace

int decode_jpeg (const char *path, std::vector<uint8_t> &rgba, int *width, int *height)
{
  try { return decode_jpeg_impl (path, rgba, width, height); }
  catch (const std::bad_alloc &) { return set_error (VFHIP_ERR_NOMEM, "%s: out of memory while decoding", path); }
}

// PNG or JPEG by the file's first bytes
int decode_image (const char *path, std::vector<uint8_t> &rgba, int *width, int *height)
{
  uint8_t sig[2] = { 0, 0 };
  FILE *f = fopen (path, "rb");
  if (!f) return set_error (VFHIP_ERR_INVALID, "cannot open %s", path);
  const size_t n = fread (sig, 1, 2, f);
  fclose (f);
  if (n == 2 && sig[0] == 0xff && sig[1] == 0xd8) return decode_jpeg (path, rgba, width, height);
  return decode_png (path, rgba, width, height);
}

}  // namespace vfhip

extern "C" int vfhip_image_decode (const char *path, uint8_t **rgba, int *width, int *height)
{
  using namespace vfhip;
  if (!path || !rgba || !width || !height) return set_error (VFHIP_ERR_INVALID, "null argument");
  std::vector<uint8_t> px;
  int rc = decode_image (path, px, width, height);
  if (rc) return rc;
  *rgba = (uint8_t *) malloc (px.size ());
  if (!*rgba) return set_error (VFHIP_ERR_NOMEM, "out of memory");
  memcpy (*rgba, px.data (), px.size ());
  return VFHIP_OK;
}
