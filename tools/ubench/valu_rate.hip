// tools/ubench/valu_rate.hip — measures the issue rate of the packed-byte VALU ops the convertscale kernel
// leans on, relative to v_add_u32 (diagnostic tool, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
template <int OP> __device__ __forceinline__ uint32_t op (uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t d;
  if (OP == 0) asm volatile ("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == 1) asm volatile ("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 2) asm volatile ("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 3) asm volatile ("v_lerp_u8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 4) asm volatile ("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(a));
  if (OP == 5) asm volatile ("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 6) asm volatile ("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == 7) asm volatile ("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == 8) asm volatile ("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 9) asm volatile ("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == 10) asm volatile ("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 11) asm volatile ("v_alignbyte_b32 %0, %1, %2, 2" : "=v"(d) : "v"(a), "v"(b));
  if (OP == 12) asm volatile ("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 13) asm volatile ("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 14) asm volatile ("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  if (OP == 15) asm volatile ("v_cvt_pk_u8_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <int OP> __global__ __launch_bounds__ (256) void k (uint32_t *out, int iters)
{
  uint32_t r[8];
  for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 7 + i;
  uint32_t b = blockIdx.x | 1, c = threadIdx.x | 3;
  for (int it = 0; it < iters; it++) {
    REP8 (r[0] = op<OP> (r[0], b, c); r[1] = op<OP> (r[1], b, c); r[2] = op<OP> (r[2], b, c); r[3] = op<OP> (r[3], b, c);
          r[4] = op<OP> (r[4], b, c); r[5] = op<OP> (r[5], b, c); r[6] = op<OP> (r[6], b, c); r[7] = op<OP> (r[7], b, c);)
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s ^= r[i];
  if (s == 0x12345) out[0] = s;
}
template <int OP> void run (const char *name, uint32_t *d)
{
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  k<OP><<<blocks, 256>>> (d, 10);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  k<OP><<<blocks, 256>>> (d, iters);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  double ops = (double) blocks * 256 * iters * 64;
  printf ("%-28s %8.3f ms  %8.1f Glane-ops/s\n", name, ms, ops / ms / 1e6);
}
int main ()
{
  uint32_t *d; hipMalloc (&d, 4096);
  run<0> ("v_add_u32", d); run<1> ("v_mad_i32_i16", d); run<12> ("v_mad_i32_i16 op_sel", d); run<2> ("v_perm_b32", d); run<3> ("v_lerp_u8", d);
  run<4> ("v_sat_pk_u8_i16", d); run<5> ("v_pk_mad_u16", d); run<9> ("v_pk_mul_lo_u16", d); run<6> ("v_and_b32", d); run<7> ("v_lshl_or_b32", d);
  run<8> ("v_mad_i32_i24", d); run<10> ("v_mad_u32_u16", d); run<11> ("v_alignbyte_b32", d); run<13> ("v_med3_i32", d);
  run<14> ("v_dot4_i32_i8", d); run<15> ("v_cvt_pk_u8_f32", d);
  return 0;
}
