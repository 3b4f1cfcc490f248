// csrc_bench/streambench.hip -> libvfhip_bench.so — streaming reference kernels for bench.py's `roofline.achievable`
// (NOT part of the product ABI: libvfhip.so does not contain or need them).  What this box's HBM gives to the simplest
// possible kernels of the same size class as the job, timed with HIP events on the caller's stream:
//   kind 0  copy            one 16-byte load + one 16-byte store per lane, lanes in address order        (read + write 1:1)
//   kind 1  mix 3:2         48 bytes in, 32 bytes out per lane, non-temporal — the job's read:write ratio (NV12 2160p in,
//                           BGRA 1080p out = 12,441,600 : 8,294,400 = 3:2) without its 2-D structure or arithmetic
//   kind 2  read only       16-byte non-temporal loads
//   kind 3  write only      16-byte stores
// MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; a lane that does MORE than one 16-byte access pair measured
// slower on this pool (tools/ubench/membw.hip, profiles/r02a_membw.txt), hence the one-access-per-lane shapes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef uint32_t v4u __attribute__ ((ext_vector_type (4)));

__global__ __launch_bounds__ (256) void sb_copy (const v4u *in, v4u *out, size_t n16)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = in[i];
}
__global__ __launch_bounds__ (256) void sb_mix32 (const uint8_t *in, uint8_t *out, size_t nlanes)
{
  const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (t >= nlanes) return;
  const uint8_t *pi = in + (size_t) blockIdx.x * 256 * 48 + threadIdx.x * 16;
  uint8_t *po = out + (size_t) blockIdx.x * 256 * 32 + threadIdx.x * 16;
  const v4u a = __builtin_nontemporal_load ((const v4u *) pi), b = __builtin_nontemporal_load ((const v4u *) (pi + 4096)),
            c = __builtin_nontemporal_load ((const v4u *) (pi + 8192));
  __builtin_nontemporal_store (a ^ c, (v4u *) po);
  __builtin_nontemporal_store (b ^ c, (v4u *) (po + 4096));
}
__global__ __launch_bounds__ (256) void sb_read (const v4u *in, uint32_t *sink, size_t n16)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n16) return;
  const v4u v = __builtin_nontemporal_load (in + i);
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u && v.x == 0x9abcdef0u) sink[0] = v.y;     // practically never: keeps the load alive
}
__global__ __launch_bounds__ (256) void sb_write (v4u *out, size_t n16)
{
  const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = v4u{ (uint32_t) i, 1u, 2u, 3u };
}

extern "C" {
// in / out: device buffers of at least in_bytes / out_bytes (kind 1 needs in_bytes >= 1.5 * out_bytes, multiples of 12 KiB / 8 KiB
// are used); returns 0 and the average ms of one pass and the bytes one pass moves (read + written), or a negative value.
int vfhip_bench_stream (int kind, const void *in, void *out, size_t in_bytes, size_t out_bytes, int warm, int reps, void *stream,
                        double *ms_per_pass, double *bytes_per_pass)
{
  if (!ms_per_pass || !bytes_per_pass || reps <= 0) return -1;
  hipStream_t s = (hipStream_t) stream;
  hipEvent_t e0, e1;
  if (hipEventCreate (&e0) != hipSuccess || hipEventCreate (&e1) != hipSuccess) return -4;
  size_t n = 0;
  double bytes = 0;
  switch (kind) {
    case 0: n = (in_bytes < out_bytes ? in_bytes : out_bytes) / 16; bytes = 32.0 * n; break;
    case 1: { size_t blocks = out_bytes / (256 * 32); const size_t bi = in_bytes / (256 * 48); if (bi < blocks) blocks = bi; n = blocks * 256; bytes = 80.0 * n; break; }
    case 2: n = in_bytes / 16; bytes = 16.0 * n; break;
    case 3: n = out_bytes / 16; bytes = 16.0 * n; break;
    default: return -1;
  }
  if (!n) return -1;
  const unsigned grid = (unsigned) ((n + 255) / 256);
  auto launch = [&] () {
    switch (kind) {
      case 0: sb_copy<<<grid, 256, 0, s>>> ((const v4u *) in, (v4u *) out, n); break;
      case 1: sb_mix32<<<grid, 256, 0, s>>> ((const uint8_t *) in, (uint8_t *) out, n); break;
      case 2: sb_read<<<grid, 256, 0, s>>> ((const v4u *) in, (uint32_t *) out, n); break;
      case 3: sb_write<<<grid, 256, 0, s>>> ((v4u *) out, n); break;
    }
  };
  for (int i = 0; i < warm; i++) launch ();
  (void) hipEventRecord (e0, s);
  for (int i = 0; i < reps; i++) launch ();
  (void) hipEventRecord (e1, s);
  if (hipEventSynchronize (e1) != hipSuccess || hipGetLastError () != hipSuccess) { (void) hipEventDestroy (e0); (void) hipEventDestroy (e1); return -4; }
  float ms = 0;
  (void) hipEventElapsedTime (&ms, e0, e1);
  (void) hipEventDestroy (e0); (void) hipEventDestroy (e1);
  *ms_per_pass = (double) ms / reps;
  *bytes_per_pass = bytes;
  return 0;
}
}
