// tools/ubench/lut_gather.hip — what does the 3D-LUT corner gather of the video filter cost per pixel, by cell layout and by WHO loads what?
// (diagnostic tool, not product code).  Random cells of a 33^3 table (the worst case the bench uses: uniform random colours).
//   A  96-byte fp32 cell (round 2's layout), each lane loads its own cell: 6 x global_load_dwordx4
//   B  48-byte fp16 cell, each lane its own cell: 3 x dwordx4
//   C  64-byte cell, each lane its own cell: 4 x dwordx4
//   D  64-byte cell, QUAD-cooperative: instruction j, the four lanes of quad q load the four 16-byte parts of pixel 4q+j's cell (one 64-byte
//      line per quad instead of one per lane), parts handed to their owner through LDS (ds_write_b128 / ds_read_b128, rotated so both are conflict-free)
//   E  as D with global_load_lds_dwordx4 (no VGPR staging, no ds_write)
// Every variant xors the loaded dwords into a checksum so the loads stay alive; prints ns per pixel-gather and the checksum (must be equal for C, D, E).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32;
typedef u32 v4u __attribute__ ((ext_vector_type (4)));
constexpr int N = 33, CELLS = N * N * N;
__device__ __forceinline__ u32 next_cell (u32 &s) { s = s * 1664525u + 1013904223u; return (u32) (((unsigned long long) (s >> 4) * (unsigned long long) CELLS) >> 28); }
__device__ __forceinline__ u32 fold (v4u v) { return v.x ^ (v.y * 3u) ^ (v.z * 5u) ^ (v.w * 7u); }

template <int PARTS, int CELL_BYTES> __global__ __launch_bounds__ (256) void k_own (const uint8_t *tab, u32 *out, int iters)
{
  u32 s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int it = 0; it < iters; it++) {
    const u32 c = next_cell (s);
    const v4u *p = reinterpret_cast<const v4u *> (tab + (size_t) c * CELL_BYTES);
    v4u v[PARTS];
#pragma unroll
    for (int k = 0; k < PARTS; k++) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < PARTS; k++) acc += fold (v[k]) * (k + 1);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <bool DMA> __global__ __launch_bounds__ (256) void k_quad (const uint8_t *tab, u32 *out, int iters)
{
  __shared__ v4u stage[4][4][64];                          // [wave][instruction j][lane]: 16 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane & 3, j_own = lane & 3, q = lane >> 2;
  u32 s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  for (int it = 0; it < iters; it++) {
    const u32 c = next_cell (s);
    const u32 base = c * 64u;
    // instruction j: this lane (quad q, position t) loads part (t - j) & 3 of pixel 4q + j's cell
#define STEP(j) { \
      const u32 bj = (u32) __builtin_amdgcn_mov_dpp ((int) base, j * 0x55, 0xf, 0xf, false);       /* quad_perm [j,j,j,j] */ \
      const uint8_t *src = tab + bj + 16u * ((t - j) & 3); \
      if (DMA) __builtin_amdgcn_global_load_lds ((const __attribute__ ((address_space (1))) void *) src, \
                                                 (__attribute__ ((address_space (3))) void *) &stage[wave][j][0], 16, 0, 0); \
      else stage[wave][j][lane] = *reinterpret_cast<const v4u *> (src); }
    STEP (0) STEP (1) STEP (2) STEP (3)
    if (DMA) asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier ();
    // this lane owns pixel 4q + j_own: part k sits in slot (j_own, 4q + ((k + j_own) & 3))
    v4u v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = stage[wave][j_own][4 * q + ((k + j_own) & 3)];
    __builtin_amdgcn_wave_barrier ();
#pragma unroll
    for (int k = 0; k < 4; k++) acc += fold (v[k]) * (k + 1);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K> static void run (const char *name, K kern, const uint8_t *tab, u32 *out, int blocks)
{
  const int iters = 200;
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  kern<<<blocks, 256>>> (tab, out, 10);
  (void) hipDeviceSynchronize ();
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    (void) hipEventRecord (e0);
    kern<<<blocks, 256>>> (tab, out, iters);
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  std::vector<u32> h ((size_t) blocks * 256);
  (void) hipMemcpy (h.data (), out, h.size () * 4, hipMemcpyDeviceToHost);
  u32 x = 0; for (u32 v : h) x ^= v;
  const double px = (double) blocks * 256 * iters;
  printf ("%-64s %8.3f ms  %7.3f ps per pixel  = %6.2f us per 1080p frame   checksum %08x  (%s)\n", name, best, best * 1e9 / px, best * 1e3 / px * 1920 * 1080, x, hipGetErrorString (hipGetLastError ()));
}

int main ()
{
  std::vector<uint8_t> h ((size_t) CELLS * 96);
  u32 s = 1; for (auto &b : h) { s = s * 1103515245u + 12345u; b = (uint8_t) (s >> 16); }
  uint8_t *tab; u32 *out;
  const int blocks = 256 * 8 * 4;
  (void) hipMalloc (&tab, h.size ()); (void) hipMalloc (&out, (size_t) blocks * 256 * 4);
  (void) hipMemcpy (tab, h.data (), h.size (), hipMemcpyHostToDevice);
  run ("A  96-byte cell, own cell per lane, 6 x dwordx4", k_own<6, 96>, tab, out, blocks);
  run ("B  48-byte cell, own cell per lane, 3 x dwordx4", k_own<3, 48>, tab, out, blocks);
  run ("C  64-byte cell, own cell per lane, 4 x dwordx4", k_own<4, 64>, tab, out, blocks);
  run ("C3 64-byte cell, own cell per lane, 3 of its 4 parts", k_own<3, 64>, tab, out, blocks);
  run ("D  64-byte cell, quad-cooperative, via VGPR + LDS", k_quad<false>, tab, out, blocks);
  run ("E  64-byte cell, quad-cooperative, global_load_lds", k_quad<true>, tab, out, blocks);
  return 0;
}
