// tools/ubench/pageable_memcpy_race.hip — is a table uploaded with a blocking hipMemcpy from PAGEABLE host memory (a std::vector) visible to a kernel
// launched right afterwards on a hipStreamNonBlocking stream?  That was libvfhip's table upload until round 2's commit 005af1b, and the suspected cause of
// three one-off gst-exact mismatches (a whole output row = one stale vertical-tap entry; four pixels; and, with every allocation poisoned, 69 % of a
// frame).  The CUDA / HIP contract only says the call returns once the pageable buffer has been STAGED; ordering is with the null stream, which a
// non-blocking stream does not join.  Diagnostic tool, not product code.
//   per trial: recycle a device table (hipFree + hipMalloc of varying sizes, like successive configures), poison it from the device, hipMemcpy a fresh
//   host table, launch a checker kernel on the non-blocking stream that counts entries still holding the poison or the PREVIOUS trial's values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_fill (uint32_t *t, int n, uint32_t v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) t[i] = v; }
__global__ void k_check (const uint32_t *t, int n, uint32_t base, unsigned long long *stale)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && t[i] != base + (uint32_t) i) atomicAdd (stale, 1ull);
}
static long run (bool in_stream, int trials, size_t max_entries)
{
  hipStream_t s; (void) hipStreamCreateWithFlags (&s, hipStreamNonBlocking);
  unsigned long long *d_stale, h_stale = 0; (void) hipMalloc (&d_stale, 8); (void) hipMemset (d_stale, 0, 8);
  uint32_t seed = 12345, bad_trials = 0;
  for (int t = 0; t < trials; t++) {
    seed = seed * 1664525u + 1013904223u;
    const int n = 16 + (int) ((seed >> 8) % max_entries);
    uint32_t *d; (void) hipMalloc (&d, (size_t) n * 4);
    k_fill<<<(n + 255) / 256, 256, 0, s>>> (d, n, 0xA5A5A5A5u);                 // what VFHIP_DEBUG_POISON does (there: hipMemset + device sync)
    (void) hipStreamSynchronize (s);
    std::vector<uint32_t> host ((size_t) n);                                      // pageable, freshly allocated: what tab_upload () handed to hipMemcpy
    const uint32_t base = seed;
    for (int i = 0; i < n; i++) host[(size_t) i] = base + (uint32_t) i;
    if (in_stream) { (void) hipMemcpyAsync (d, host.data (), (size_t) n * 4, hipMemcpyHostToDevice, s); (void) hipStreamSynchronize (s); }
    else (void) hipMemcpy (d, host.data (), (size_t) n * 4, hipMemcpyHostToDevice);
    unsigned long long before; (void) hipMemcpy (&before, d_stale, 8, hipMemcpyDeviceToHost);
    k_check<<<(n + 255) / 256, 256, 0, s>>> (d, n, base, d_stale);
    (void) hipStreamSynchronize (s);
    (void) hipMemcpy (&h_stale, d_stale, 8, hipMemcpyDeviceToHost);
    if (h_stale != before) bad_trials++;
    (void) hipFree (d);
  }
  printf ("%-62s trials %d (tables of 16..%zu entries)   trials with stale entries %u   stale entries %llu\n",
          in_stream ? "hipMemcpyAsync in the kernel's stream + hipStreamSynchronize" : "blocking hipMemcpy (null stream), kernel on a non-blocking stream", trials, max_entries, bad_trials, h_stale);
  (void) hipFree (d_stale); (void) hipStreamDestroy (s);
  return (long) bad_trials;
}
int main ()
{
  run (false, 20000, 4096);
  run (false, 5000, 1 << 18);
  run (true, 20000, 4096);
  run (true, 5000, 1 << 18);
  return 0;
}
