// tools/ubench/bperm_rate.hip — ds_bpermute_b32 throughput on gfx950 next to a gather of the same shape (diagnostic tool, not product code):
// (a) 8 independent bpermute chains per lane, (b) 2-byte gathers at a 3-byte lane stride, (c) 8-byte gathers at a 6-byte lane stride,
// (d) coalesced dword loads — wave-instructions per microsecond per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__ (256) void k_bperm (uint32_t *out, int iters)
{
  uint32_t r[8];
  for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 7 + i;
  const int idx = ((threadIdx.x * 3) & 63) * 4;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (uint32_t) __builtin_amdgcn_ds_bpermute (idx, (int) r[i]);
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= r[i];
  if (s == 0x12345) out[0] = s;
}
template <int MODE> __global__ __launch_bounds__ (256) void k_gather (const uint8_t *in, uint32_t *out, int iters, int rowbytes)
{
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  uint32_t acc = 0;
  const uint8_t *base = in + (size_t) (wave % 4096) * rowbytes;
  for (int it = 0; it < iters; it++) {
    const uint8_t *row = base + (size_t) (it & 7) * 512;
    if (MODE == 0) acc += *reinterpret_cast<const uint16_t *> (row + 3 * lane + (lane & 1));
    else if (MODE == 1) { typedef uint2 __attribute__ ((aligned (2))) u2a; const uint2 v = *reinterpret_cast<const u2a *> (row + 6 * lane); acc += v.x ^ v.y; }
    else acc += *reinterpret_cast<const uint32_t *> (row + 4 * lane);
  }
  if (acc == 0x12345) out[0] = acc;
}
int main ()
{
  uint32_t *d; uint8_t *in;
  (void) hipMalloc (&d, 4096); (void) hipMalloc (&in, (size_t) 4096 * 4096 + 8192); (void) hipMemset (in, 1, (size_t) 4096 * 4096 + 8192);
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  const int blocks = 256 * 8, iters = 2000;
  float ms;
  k_bperm<<<blocks, 256>>> (d, 10); (void) hipDeviceSynchronize ();
  (void) hipEventRecord (e0); k_bperm<<<blocks, 256>>> (d, iters); (void) hipEventRecord (e1); (void) hipEventSynchronize (e1); (void) hipEventElapsedTime (&ms, e0, e1);
  printf ("ds_bpermute_b32         %8.3f ms  %8.1f wave-instr/us/CU  (%.2f cycles per instr per CU at 2.4 GHz)\n", ms, (double) blocks * 4 * iters * 8 / ms / 1e3 / 256, 2400.0 / ((double) blocks * 4 * iters * 8 / ms / 1e3 / 256));
#define G(M, NAME) k_gather<M><<<blocks, 256>>> (in, d, 10, 4096); (void) hipDeviceSynchronize (); (void) hipEventRecord (e0); k_gather<M><<<blocks, 256>>> (in, d, iters, 4096); (void) hipEventRecord (e1); \
  (void) hipEventSynchronize (e1); (void) hipEventElapsedTime (&ms, e0, e1); \
  printf ("%-22s  %8.3f ms  %8.1f wave-instr/us/CU  (%.2f cycles per instr per CU at 2.4 GHz)\n", NAME, ms, (double) blocks * 4 * iters / ms / 1e3 / 256, 2400.0 / ((double) blocks * 4 * iters / ms / 1e3 / 256));
  G (0, "u16 gather, stride 3") G (1, "b64 gather, stride 6") G (2, "dword coalesced")
  return 0;
}
