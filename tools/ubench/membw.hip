// tools/ubench/membw.hip — what does this box's HBM give to (a) plain streaming copies in several shapes and (b) the
// access pattern of the 2:1 NV12 -> BGRA kernel in several mappings?  Diagnostic tool, not product code: the arithmetic
// is replaced by XORs so only the memory system is measured.  One line per variant; every variant runs `reps`
// back-to-back launches after a 1 s pre-conditioning phase (sustained clock, not boost).
//   ./membw [frames=128] [reps=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));
typedef uint32_t v2u __attribute__ ((ext_vector_type (2)));

template <bool NT> __device__ __forceinline__ v4u ld16 (const void *p)
{
  if (NT) return __builtin_nontemporal_load (reinterpret_cast<const v4u *> (p));
  return *reinterpret_cast<const v4u *> (p);
}
template <bool NT> __device__ __forceinline__ v2u ld8 (const void *p)
{
  if (NT) return __builtin_nontemporal_load (reinterpret_cast<const v2u *> (p));
  return *reinterpret_cast<const v2u *> (p);
}
template <bool NT> __device__ __forceinline__ void st16 (void *p, v4u v)
{
  if (NT) __builtin_nontemporal_store (v, reinterpret_cast<v4u *> (p));
  else *reinterpret_cast<v4u *> (p) = v;
}

// ---------------------------------------------------------------------------------------------- copies
// U x 16 bytes per lane, block-contiguous chunks of 256*U*16 bytes, loads first, then stores
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__ (256) void k_copy (const uint8_t *in, uint8_t *out, size_t n16)
{
  const size_t base = ((size_t) blockIdx.x * U) * 256 + threadIdx.x;
  v4u v[U];
#pragma unroll
  for (int u = 0; u < U; u++) { const size_t i = base + (size_t) u * 256; v[u] = i < n16 ? ld16<NTL> (in + i * 16) : v4u{0, 0, 0, 0}; }
#pragma unroll
  for (int u = 0; u < U; u++) { const size_t i = base + (size_t) u * 256; if (i < n16) st16<NTS> (out + i * 16, v[u]); }
}
// persistent grid-stride copy, U x 16 bytes in flight per lane
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__ (256) void k_copy_gs (const uint8_t *in, uint8_t *out, size_t n16)
{
  const size_t stride = (size_t) gridDim.x * 256 * U;
  for (size_t b = ((size_t) blockIdx.x * U) * 256 + threadIdx.x; b < n16; b += stride) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = b + (size_t) u * 256; v[u] = i < n16 ? ld16<NTL> (in + i * 16) : v4u{0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = b + (size_t) u * 256; if (i < n16) st16<NTS> (out + i * 16, v[u]); }
  }
}
template <int U, bool NT>
__global__ __launch_bounds__ (256) void k_read (const uint8_t *in, uint32_t *sink, size_t n16)
{
  const size_t base = ((size_t) blockIdx.x * U) * 256 + threadIdx.x;
  uint32_t acc = 0;
#pragma unroll
  for (int u = 0; u < U; u++) { const size_t i = base + (size_t) u * 256; if (i < n16) { const v4u v = ld16<NT> (in + i * 16); acc ^= v.x ^ v.y ^ v.z ^ v.w; } }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__ (256) void k_write (uint8_t *out, size_t n16)
{
  const size_t base = ((size_t) blockIdx.x * U) * 256 + threadIdx.x;
  const v4u v = { (uint32_t) base, 1, 2, 3 };
#pragma unroll
  for (int u = 0; u < U; u++) { const size_t i = base + (size_t) u * 256; if (i < n16) st16<NT> (out + i * 16, v); }
}
// flat 3:2 mix (read 48 B, write 32 B per lane): the job's read/write ratio without its 2-D structure
template <bool NTL, bool NTS>
__global__ __launch_bounds__ (256) void k_mix32 (const uint8_t *in, uint8_t *out, size_t nlanes)
{
  const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (t >= nlanes) return;
  const size_t bi = (size_t) blockIdx.x * 256 * 48 + threadIdx.x * 16, bo = (size_t) blockIdx.x * 256 * 32 + threadIdx.x * 16;
  const v4u a = ld16<NTL> (in + bi), b = ld16<NTL> (in + bi + 4096), c = ld16<NTL> (in + bi + 8192);
  st16<NTS> (out + bo, a ^ c); st16<NTS> (out + bo + 4096, b ^ c);
}

// ---------------------------------------------------------------------------------------------- the 2:1 pattern
// P8: lane = 4 output pixels (8-byte luma loads x2, 8-byte chroma load, 16-byte store per output row), strips of ROWS
// output rows, lanes flattened over (strip, column group); PF = how many rows ahead the loads are issued (0, 1, 2).
template <int ROWS, int PF, bool NTL, bool NTS>
__global__ __launch_bounds__ (256) void k_p8 (const uint8_t *in, uint8_t *out, int ow, int oh, size_t in_pitch, size_t out_pitch, int bpf)
{
  const int cgpr = ow >> 2, strips = (oh + ROWS - 1) / ROWS;
  const int frame = blockIdx.x / bpf, t = (blockIdx.x % bpf) * 256 + threadIdx.x;
  if (t >= cgpr * strips) return;
  const int strip = t / cgpr, cg = t - strip * cgpr, y0 = strip * ROWS, y1 = min (y0 + ROWS, oh);
  const uint32_t ys = 2 * ow;
  const uint8_t *yp = in + (size_t) frame * in_pitch, *uvp = yp + (size_t) ys * (2 * oh);
  uint8_t *op = out + (size_t) frame * out_pitch;
  v2u a[PF + 1], b[PF + 1], c[PF + 1];
  auto load = [&] (int y, int s) {
    const int yy = min (y, y1 - 1);
    a[s] = ld8<NTL> (yp + (uint32_t) (2 * yy) * ys + 8u * cg);
    b[s] = ld8<NTL> (yp + (uint32_t) (2 * yy + 1) * ys + 8u * cg);
    c[s] = ld8<NTL> (uvp + (uint32_t) yy * ys + 8u * cg);
  };
#pragma unroll
  for (int s = 0; s < PF; s++) load (y0 + s, s);
  uint32_t acc = 0;
#pragma unroll 1
  for (int y = y0; y < y1; y += PF + 1) {
#pragma unroll
    for (int s = 0; s <= PF; s++) {
      if (y + s >= y1) break;
      load (y + s + PF, (s + PF) % (PF + 1));
      const v2u aa = a[s], bb = b[s], cc = c[s];
      acc ^= cc.y;
      st16<NTS> (op + (uint32_t) (y + s) * (4u * ow) + 16u * cg, v4u{aa.x ^ acc, aa.y ^ cc.x, bb.x ^ cc.y, bb.y});
    }
  }
}
// P16: lane = 8 output pixels (16-byte loads, two 16-byte stores)
template <int ROWS, int PF, bool NTL, bool NTS>
__global__ __launch_bounds__ (256) void k_p16 (const uint8_t *in, uint8_t *out, int ow, int oh, size_t in_pitch, size_t out_pitch, int bpf)
{
  const int cgpr = ow >> 3, strips = (oh + ROWS - 1) / ROWS;
  const int frame = blockIdx.x / bpf, t = (blockIdx.x % bpf) * 256 + threadIdx.x;
  if (t >= cgpr * strips) return;
  const int strip = t / cgpr, cg = t - strip * cgpr, y0 = strip * ROWS, y1 = min (y0 + ROWS, oh);
  const uint32_t ys = 2 * ow;
  const uint8_t *yp = in + (size_t) frame * in_pitch, *uvp = yp + (size_t) ys * (2 * oh);
  uint8_t *op = out + (size_t) frame * out_pitch;
  v4u a[PF + 1], b[PF + 1], c[PF + 1];
  auto load = [&] (int y, int s) {
    const int yy = min (y, y1 - 1);
    a[s] = ld16<NTL> (yp + (uint32_t) (2 * yy) * ys + 16u * cg);
    b[s] = ld16<NTL> (yp + (uint32_t) (2 * yy + 1) * ys + 16u * cg);
    c[s] = ld16<NTL> (uvp + (uint32_t) yy * ys + 16u * cg);
  };
#pragma unroll
  for (int s = 0; s < PF; s++) load (y0 + s, s);
#pragma unroll 1
  for (int y = y0; y < y1; y += PF + 1) {
#pragma unroll
    for (int s = 0; s <= PF; s++) {
      if (y + s >= y1) break;
      load (y + s + PF, (s + PF) % (PF + 1));
      const v4u aa = a[s], bb = b[s], cc = c[s];
      uint8_t *o = op + (uint32_t) (y + s) * (4u * ow) + 32u * cg;
      st16<NTS> (o, v4u{aa.x ^ cc.x, aa.y ^ cc.y, bb.x, bb.y});
      st16<NTS> (o + 16, v4u{aa.z ^ cc.z, aa.w ^ cc.w, bb.z, bb.w});
    }
  }
}
// PR: whole-row blocks.  One block owns ROWS consecutive output rows over the full width (240 of 256 lanes active at
// 1920 columns, 8 output pixels per lane): every block streams contiguous memory like a copy.
template <int ROWS, int PF, bool NTL, bool NTS>
__global__ __launch_bounds__ (256) void k_prow (const uint8_t *in, uint8_t *out, int ow, int oh, size_t in_pitch, size_t out_pitch, int bpf)
{
  const int frame = blockIdx.x / bpf, strip = blockIdx.x % bpf;
  const int cg = threadIdx.x;
  if (cg >= (ow >> 3)) return;
  const int y0 = strip * ROWS, y1 = min (y0 + ROWS, oh);
  const uint32_t ys = 2 * ow;
  const uint8_t *yp = in + (size_t) frame * in_pitch, *uvp = yp + (size_t) ys * (2 * oh);
  uint8_t *op = out + (size_t) frame * out_pitch;
  v4u a[PF + 1], b[PF + 1], c[PF + 1];
  auto load = [&] (int y, int s) {
    const int yy = min (y, y1 - 1);
    a[s] = ld16<NTL> (yp + (uint32_t) (2 * yy) * ys + 16u * cg);
    b[s] = ld16<NTL> (yp + (uint32_t) (2 * yy + 1) * ys + 16u * cg);
    c[s] = ld16<NTL> (uvp + (uint32_t) yy * ys + 16u * cg);
  };
#pragma unroll
  for (int s = 0; s < PF; s++) load (y0 + s, s);
#pragma unroll 1
  for (int y = y0; y < y1; y += PF + 1) {
#pragma unroll
    for (int s = 0; s <= PF; s++) {
      if (y + s >= y1) break;
      load (y + s + PF, (s + PF) % (PF + 1));
      const v4u aa = a[s], bb = b[s], cc = c[s];
      uint8_t *o = op + (uint32_t) (y + s) * (4u * ow) + 32u * cg;
      st16<NTS> (o, v4u{aa.x ^ cc.x, aa.y ^ cc.y, bb.x, bb.y});
      st16<NTS> (o + 16, v4u{aa.z ^ cc.z, aa.w ^ cc.w, bb.z, bb.w});
    }
  }
}

// ---------------------------------------------------------------------------------------------- harness
static hipEvent_t e0, e1;
static double timed (int reps, const std::function<void ()> &launch)
{
  for (int i = 0; i < 5; i++) launch ();
  (void) hipDeviceSynchronize ();
  (void) hipEventRecord (e0);
  for (int i = 0; i < reps; i++) launch ();
  (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
  float ms; (void) hipEventElapsedTime (&ms, e0, e1);
  hipError_t err = hipGetLastError ();
  if (err != hipSuccess) { printf ("HIP error: %s\n", hipGetErrorString (err)); exit (1); }
  return ms / reps;
}
static void report (const char *name, double ms, double bytes)
{
  printf ("%-58s %8.4f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush (stdout);
}

int main (int argc, char **argv)
{
  const int F = argc > 1 ? atoi (argv[1]) : 128, reps = argc > 2 ? atoi (argv[2]) : 200;
  const int ow = 1920, oh = 1080;
  const size_t in_pitch0 = (size_t) 3840 * 2160 * 3 / 2, out_pitch0 = (size_t) ow * oh * 4;
  const size_t pad = 1 << 20;
  uint8_t *in, *out; uint32_t *sink;
  (void) hipMalloc (&in, (in_pitch0 + pad) * F + 4096); (void) hipMalloc (&out, (out_pitch0 + pad) * F + 4096); (void) hipMalloc (&sink, 4096);
  // random-ish bytes (bit toggling costs power: constant data flatters the result)
  {
    const size_t n = (in_pitch0 + pad) * F;
    uint32_t *h = (uint32_t *) malloc (64 << 20);
    uint32_t s = 0x9E3779B9u;
    for (size_t i = 0; i < (64u << 20) / 4; i++) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; h[i] = s; }
    for (size_t o = 0; o < n; o += 64u << 20) (void) hipMemcpy (in + o, h, n - o < (64u << 20) ? n - o : (64u << 20), hipMemcpyHostToDevice);
    free (h);
  }
  (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  const size_t nin = in_pitch0 * F, nout = out_pitch0 * F;

  // pre-conditioning: ~1 s of copies
  {
    const size_t n16 = nout / 16;
    for (int i = 0; i < 1500; i++) k_copy<4, false, false><<<(unsigned) ((n16 + 1023) / 1024), 256>>> (in, out, n16);
    (void) hipDeviceSynchronize ();
  }
  char name[160];
#define COPY(U, NTL, NTS) { const size_t n16 = nout / 16; const unsigned g = (unsigned) ((n16 + 256 * U - 1) / (256 * U)); \
    const double ms = timed (reps, [&] () { k_copy<U, NTL, NTS><<<g, 256>>> (in, out, n16); }); \
    snprintf (name, sizeof name, "copy %zu MB U=%d ntl=%d nts=%d (read+write)", nout >> 20, U, NTL, NTS); report (name, ms, 2.0 * nout); }
  COPY (1, false, false) COPY (2, false, false) COPY (4, false, false) COPY (8, false, false)
  COPY (4, true, false) COPY (4, false, true) COPY (4, true, true) COPY (8, true, true)
#define COPYGS(U, NTL, NTS, BLK) { const size_t n16 = nout / 16; \
    const double ms = timed (reps, [&] () { k_copy_gs<U, NTL, NTS><<<BLK, 256>>> (in, out, n16); }); \
    snprintf (name, sizeof name, "copy grid-stride blocks=%d U=%d ntl=%d nts=%d", BLK, U, NTL, NTS); report (name, ms, 2.0 * nout); }
  COPYGS (4, false, false, 2048) COPYGS (4, false, false, 1024) COPYGS (8, false, false, 1024) COPYGS (4, true, true, 2048) COPYGS (1, false, false, 2048)
  COPYGS (2, false, false, 4096)
#define READ(U, NT) { const size_t n16 = nin / 16; const unsigned g = (unsigned) ((n16 + 256 * U - 1) / (256 * U)); \
    const double ms = timed (reps, [&] () { k_read<U, NT><<<g, 256>>> (in, sink, n16); }); \
    snprintf (name, sizeof name, "read-only %zu MB U=%d nt=%d", nin >> 20, U, NT); report (name, ms, 1.0 * nin); }
  READ (1, false) READ (4, false) READ (8, false) READ (4, true)
#define WRITE(U, NT) { const size_t n16 = nout / 16; const unsigned g = (unsigned) ((n16 + 256 * U - 1) / (256 * U)); \
    const double ms = timed (reps, [&] () { k_write<U, NT><<<g, 256>>> (out, n16); }); \
    snprintf (name, sizeof name, "write-only %zu MB U=%d nt=%d", nout >> 20, U, NT); report (name, ms, 1.0 * nout); }
  WRITE (1, false) WRITE (4, false) WRITE (4, true)
  {
    const size_t nl = nout / 32;
    const unsigned g = (unsigned) ((nl + 255) / 256);
    double ms = timed (reps, [&] () { k_mix32<false, false><<<g, 256>>> (in, out, nl); });
    report ("flat 3:2 mix (48 B in, 32 B out per lane)", ms, 80.0 * nl);
    ms = timed (reps, [&] () { k_mix32<false, true><<<g, 256>>> (in, out, nl); });
    report ("flat 3:2 mix, nt stores", ms, 80.0 * nl);
    ms = timed (reps, [&] () { k_mix32<true, true><<<g, 256>>> (in, out, nl); });
    report ("flat 3:2 mix, nt loads + stores", ms, 80.0 * nl);
  }

  const double alg = (double) (in_pitch0 + out_pitch0) * F;
#define P8(ROWS, PF, NTL, NTS, IP, OP) { const int strips = (oh + ROWS - 1) / ROWS, bpf = ((ow / 4) * strips + 255) / 256; \
    const double ms = timed (reps, [&] () { k_p8<ROWS, PF, NTL, NTS><<<bpf * F, 256>>> (in, out, ow, oh, IP, OP, bpf); }); \
    snprintf (name, sizeof name, "pattern  8B rows=%2d pf=%d ntl=%d nts=%d pitch_pad=%zu", ROWS, PF, NTL, NTS, (size_t) (IP) - in_pitch0); report (name, ms, alg); }
#define P16(ROWS, PF, NTL, NTS, IP, OP) { const int strips = (oh + ROWS - 1) / ROWS, bpf = ((ow / 8) * strips + 255) / 256; \
    const double ms = timed (reps, [&] () { k_p16<ROWS, PF, NTL, NTS><<<bpf * F, 256>>> (in, out, ow, oh, IP, OP, bpf); }); \
    snprintf (name, sizeof name, "pattern 16B rows=%2d pf=%d ntl=%d nts=%d pitch_pad=%zu", ROWS, PF, NTL, NTS, (size_t) (IP) - in_pitch0); report (name, ms, alg); }
#define PROW(ROWS, PF, NTL, NTS, IP, OP) { const int bpf = (oh + ROWS - 1) / ROWS; \
    const double ms = timed (reps, [&] () { k_prow<ROWS, PF, NTL, NTS><<<bpf * F, 256>>> (in, out, ow, oh, IP, OP, bpf); }); \
    snprintf (name, sizeof name, "whole-row 16B rows=%2d pf=%d ntl=%d nts=%d pitch_pad=%zu", ROWS, PF, NTL, NTS, (size_t) (IP) - in_pitch0); report (name, ms, alg); }
  P8 (16, 0, false, true, in_pitch0, out_pitch0) P8 (16, 1, false, true, in_pitch0, out_pitch0) P8 (16, 2, false, true, in_pitch0, out_pitch0)
  P8 (8, 1, false, true, in_pitch0, out_pitch0) P8 (4, 1, false, true, in_pitch0, out_pitch0) P8 (4, 0, false, true, in_pitch0, out_pitch0)
  P8 (16, 1, false, false, in_pitch0, out_pitch0) P8 (16, 1, true, true, in_pitch0, out_pitch0)
  P8 (16, 1, false, true, in_pitch0 + 4096 + 256, out_pitch0 + 4096 + 256) P8 (16, 1, false, true, in_pitch0 + 65536 + 256, out_pitch0 + 65536 + 256)
  P16 (16, 0, false, true, in_pitch0, out_pitch0) P16 (16, 1, false, true, in_pitch0, out_pitch0) P16 (16, 2, false, true, in_pitch0, out_pitch0)
  P16 (8, 1, false, true, in_pitch0, out_pitch0) P16 (8, 2, false, true, in_pitch0, out_pitch0) P16 (4, 1, false, true, in_pitch0, out_pitch0)
  P16 (16, 1, true, true, in_pitch0, out_pitch0) P16 (16, 1, false, false, in_pitch0, out_pitch0)
  PROW (16, 1, false, true, in_pitch0, out_pitch0) PROW (8, 1, false, true, in_pitch0, out_pitch0) PROW (8, 2, false, true, in_pitch0, out_pitch0)
  PROW (4, 1, false, true, in_pitch0, out_pitch0) PROW (4, 3, false, true, in_pitch0, out_pitch0) PROW (2, 1, false, true, in_pitch0, out_pitch0)
  PROW (8, 1, true, true, in_pitch0, out_pitch0)
  return 0;
}
