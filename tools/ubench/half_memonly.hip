// tools/ubench/half_memonly.hip — the memory access pattern of k_cs_nv12_half (per lane and row: 2 x 8-byte luma
// loads, one 8-byte + one 2-byte chroma load, one 16-byte store; strips of ROWS rows) with the arithmetic
// replaced by XORs: the bandwidth ceiling of the pattern itself (diagnostic tool, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int ROWS>
__global__ __launch_bounds__ (256) void k (const uint8_t *in, uint8_t *out, int ow, int oh, size_t in_pitch, size_t out_pitch)
{
  const int cgpr = ow >> 2, strips = (oh + ROWS - 1) / ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= cgpr * strips) return;
  const int strip = t / cgpr, cg = t - strip * cgpr, y0 = strip * ROWS;
  const int ys = 2 * ow;
  const uint8_t *yp = in + (size_t) blockIdx.y * in_pitch, *uvp = yp + (size_t) ys * (2 * oh);
  uint8_t *op = out + (size_t) blockIdx.y * out_pitch;
  uint32_t acc = 0;
  for (int y = y0; y < min (y0 + ROWS, oh); y++) {
    const uint2 c = *reinterpret_cast<const uint2 *> (uvp + (size_t) min (y + 1, oh - 1) * ys + 8 * cg);
    const uint32_t r = cg == cgpr - 1 ? 0 : *reinterpret_cast<const uint16_t *> (uvp + (size_t) min (y + 1, oh - 1) * ys + 8 * cg + 8);
    const uint2 a = *reinterpret_cast<const uint2 *> (yp + (size_t) (2 * y) * ys + 8 * cg);
    const uint2 b = *reinterpret_cast<const uint2 *> (yp + (size_t) (2 * y + 1) * ys + 8 * cg);
    acc ^= r;
    *reinterpret_cast<uint4 *> (op + (size_t) y * (4 * ow) + 16 * cg) = make_uint4 (a.x ^ acc, a.y ^ c.x, b.x ^ c.y, b.y);
  }
}
// the same traffic with 16-byte loads: one lane = 8 output pixels (2 x 16-byte luma loads, one 16-byte chroma load, two 16-byte stores)
template <int ROWS>
__global__ __launch_bounds__ (256) void k16 (const uint8_t *in, uint8_t *out, int ow, int oh, size_t in_pitch, size_t out_pitch)
{
  const int cgpr = ow >> 3, strips = (oh + ROWS - 1) / ROWS;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= cgpr * strips) return;
  const int strip = t / cgpr, cg = t - strip * cgpr, y0 = strip * ROWS;
  const int ys = 2 * ow;
  const uint8_t *yp = in + (size_t) blockIdx.y * in_pitch, *uvp = yp + (size_t) ys * (2 * oh);
  uint8_t *op = out + (size_t) blockIdx.y * out_pitch;
  for (int y = y0; y < min (y0 + ROWS, oh); y++) {
    const uint4 c = *reinterpret_cast<const uint4 *> (uvp + (size_t) min (y + 1, oh - 1) * ys + 16 * cg);
    const uint4 a = *reinterpret_cast<const uint4 *> (yp + (size_t) (2 * y) * ys + 16 * cg);
    const uint4 b = *reinterpret_cast<const uint4 *> (yp + (size_t) (2 * y + 1) * ys + 16 * cg);
    uint4 *o = reinterpret_cast<uint4 *> (op + (size_t) y * (4 * ow) + 32 * cg);
    o[0] = make_uint4 (a.x ^ c.x, a.y ^ c.y, b.x, b.y);
    o[1] = make_uint4 (a.z ^ c.z, a.w ^ c.w, b.z, b.w);
  }
}
__global__ __launch_bounds__ (256) void kcopy (const uint4 *in, uint4 *out, size_t n)
{
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) out[i] = in[i];
}
int main (int argc, char **argv)
{
  const int ow = 1920, oh = 1080, F = argc > 1 ? atoi (argv[1]) : 128;
  const size_t in_pitch = (size_t) 3840 * 2160 * 3 / 2, out_pitch = (size_t) ow * oh * 4;
  uint8_t *in, *out;
  (void) hipMalloc (&in, in_pitch * F + 256); (void) hipMalloc (&out, out_pitch * F);
  (void) hipMemset (in, 0x5a, in_pitch * F);
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  const int cgpr = ow / 4;
  auto run = [&] (int rows) {
    const int strips = (oh + rows - 1) / rows;
    dim3 grid ((cgpr * strips + 255) / 256, F);
    auto launch = [&] () { if (rows == 16) k<16><<<grid, 256>>> (in, out, ow, oh, in_pitch, out_pitch); else if (rows == 8) k<8><<<grid, 256>>> (in, out, ow, oh, in_pitch, out_pitch); else k<4><<<grid, 256>>> (in, out, ow, oh, in_pitch, out_pitch); };
    for (int it = 0; it < 3; it++) launch ();
    (void) hipDeviceSynchronize ();
    (void) hipEventRecord (e0);
    const int N = 20;
    for (int it = 0; it < N; it++) launch ();
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1); ms /= N;
    printf ("pattern rows=%2d frames=%d  %.4f ms/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)  %.0f frames/s\n", rows, F, ms,
            (double) (in_pitch + out_pitch) * F / ms / 1e6, (double) (in_pitch + out_pitch) * F / ms / 1e6 / 80.0, F / ms * 1e3);
  };
  const int N_LONG = argc > 2 ? atoi (argv[2]) : 20;
  run (16); run (8); run (4);
  auto run16 = [&] (int rows) {
    const int strips = (oh + rows - 1) / rows;
    dim3 grid (((ow / 8) * strips + 255) / 256, F);
    auto launch = [&] () { if (rows == 16) k16<16><<<grid, 256>>> (in, out, ow, oh, in_pitch, out_pitch); else k16<8><<<grid, 256>>> (in, out, ow, oh, in_pitch, out_pitch); };
    for (int it = 0; it < 3; it++) launch ();
    (void) hipDeviceSynchronize ();
    (void) hipEventRecord (e0);
    for (int it = 0; it < N_LONG; it++) launch ();
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1); ms /= N_LONG;
    printf ("pattern16 rows=%2d frames=%d launches=%d  %.4f ms/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", rows, F, N_LONG, ms,
            (double) (in_pitch + out_pitch) * F / ms / 1e6, (double) (in_pitch + out_pitch) * F / ms / 1e6 / 80.0);
  };
  run16 (16); run16 (8);
  // sustained: the 8-byte pattern again over N_LONG launches (short runs ride the boost clock)
  {
    const int strips = (oh + 15) / 16;
    dim3 grid ((cgpr * strips + 255) / 256, F);
    (void) hipEventRecord (e0);
    for (int it = 0; it < N_LONG; it++) k<16><<<grid, 256>>> (in, out, ow, oh, in_pitch, out_pitch);
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1); ms /= N_LONG;
    printf ("pattern rows=16 sustained over %d launches: %.4f ms/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", N_LONG, ms,
            (double) (in_pitch + out_pitch) * F / ms / 1e6, (double) (in_pitch + out_pitch) * F / ms / 1e6 / 80.0);
  }
  {
    const size_t n = out_pitch * F / 16;     // copy out_pitch*F bytes from in to out: read + write
    for (int it = 0; it < 3; it++) kcopy<<<256 * 8, 256>>> ((const uint4 *) in, (uint4 *) out, n);
    (void) hipDeviceSynchronize ();
    (void) hipEventRecord (e0);
    for (int it = 0; it < 20; it++) kcopy<<<256 * 8, 256>>> ((const uint4 *) in, (uint4 *) out, n);
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1); ms /= 20;
    printf ("float4 copy %zu MB: %.4f ms  %.1f GB/s (read+write)\n", n * 16 >> 20, ms, 2.0 * n * 16 / ms / 1e6);
  }
  return 0;
}
