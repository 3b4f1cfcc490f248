// tools/ubench/valu_mix.hip — do VALU opcodes of different kinds overlap when they alternate in one wave's instruction stream?  valu_occ.hip measured
// "3 v_fmac : 1 v_cvt_f32_ubyte" at 2.13 cycles per instruction, as if the half-rate conversion cost nothing extra: this tool measures pairs (A alone,
// B alone, A B A B ...) of the opcodes the hot kernels are made of.  8 independent chains per lane, 8 waves per SIMD.  Diagnostic tool, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef unsigned long long u64;
template <int OP> __device__ __forceinline__ void op (u32 &r, u64 &rr, u32 x, u32 y, u64 xx)
{
  if (OP == 0) asm volatile ("v_mad_i32_i16 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
  else if (OP == 1) asm volatile ("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
  else if (OP == 2) asm volatile ("v_lerp_u8 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
  else if (OP == 3) asm volatile ("v_sat_pk_u8_i16 %0, %0" : "+v"(r));
  else if (OP == 4) asm volatile ("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
  else if (OP == 5) asm volatile ("v_and_b32 %0, 0xffff00ff, %0" : "+v"(r));
  else if (OP == 6) asm volatile ("v_add_u32 %0, %1, %0" : "+v"(r) : "v"(x));
  else if (OP == 7) asm volatile ("v_cvt_f32_ubyte1 %0, %0" : "+v"(r));
  else if (OP == 8) asm volatile ("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
  else if (OP == 9) asm volatile ("v_rndne_f32 %0, %0" : "+v"(r));
  else if (OP == 10) asm volatile ("v_pk_mul_f32 %0, %0, %1" : "+v"(rr) : "v"(xx));
  else if (OP == 11) asm volatile ("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(x));
  else if (OP == 12) asm volatile ("v_cvt_pk_u8_f32 %0, %0, 1, %1" : "+v"(r) : "v"(x));
  else if (OP == 13) asm volatile ("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r) : "v"(x));
  else if (OP == 14) asm volatile ("v_xor_b32 %0, 0x80808080, %0" : "+v"(r));
  else if (OP == 15) asm volatile ("v_exp_f32 %0, %0" : "+v"(r));
  else if (OP == 16) asm volatile ("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(rr) : "v"(xx));
  else if (OP == 17) asm volatile ("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(r) : "v"(x), "v"(y));
}
template <int A, int B, int NA, int NB> __global__ __launch_bounds__ (512) void k (u32 *out, int iters)
{
  u32 r[8]; u64 rr[8];
  for (int i = 0; i < 8; i++) { r[i] = 0x3f000000u + threadIdx.x * 7 + i; rr[i] = ((u64) r[i] << 32) | r[i]; }
  const u32 x = 0x3f800000u | blockIdx.x, y = 0x3e000000u | threadIdx.x;
  const u64 xx = ((u64) x << 32) | y;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
#pragma unroll
      for (int a = 0; a < NA; a++)
#pragma unroll
        for (int i = 0; i < 8; i++) op<A> (r[i], rr[i], x, y, xx);
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int i = 0; i < 8; i++) op<B> (r[i], rr[i], x, y, xx);
    }
  }
  u32 s = 0;
  for (int i = 0; i < 8; i++) s += r[i] + (u32) rr[i];
  if (s == 0x12345u) out[0] = s;
}
// FINE: A and B alternate instruction by instruction (chain i takes A, chain i + 1 takes B, roles swap every round)
template <int A, int B> __global__ __launch_bounds__ (512) void kf (u32 *out, int iters)
{
  u32 r[8]; u64 rr[8];
  for (int i = 0; i < 8; i++) { r[i] = 0x3f000000u + threadIdx.x * 7 + i; rr[i] = ((u64) r[i] << 32) | r[i]; }
  const u32 x = 0x3f800000u | blockIdx.x, y = 0x3e000000u | threadIdx.x;
  const u64 xx = ((u64) x << 32) | y;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int i = 0; i < 8; i += 2) { op<A> (r[i], rr[i], x, y, xx); op<B> (r[i + 1], rr[i + 1], x, y, xx); }
  }
  u32 s = 0;
  for (int i = 0; i < 8; i++) s += r[i] + (u32) rr[i];
  if (s == 0x12345u) out[0] = s;
}
static float timed (void (*launch) (u32 *, int), u32 *d, int iters)
{
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  launch (d, 10); (void) hipDeviceSynchronize ();
  float best = 1e9f;
  for (int t = 0; t < 3; t++) {
    (void) hipEventRecord (e0); launch (d, iters); (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1); best = ms < best ? ms : best;
  }
  return best;
}
static const int BLOCKS = 256 * 4, ITERS = 1000;       // 512-lane workgroups, 4 per CU: 8 waves per SIMD
template <int A, int B> void pair (const char *na, const char *nb, u32 *d, double ref)
{
  const float ta = timed ([] (u32 *p, int n) { k<A, A, 1, 0><<<BLOCKS, 512>>> (p, n); }, d, ITERS);
  const float tb = timed ([] (u32 *p, int n) { k<B, B, 1, 0><<<BLOCKS, 512>>> (p, n); }, d, ITERS);
  const float tg = timed ([] (u32 *p, int n) { k<A, B, 1, 1><<<BLOCKS, 512>>> (p, n); }, d, ITERS);      // groups of 8 A, 8 B
  const float tf = timed ([] (u32 *p, int n) { kf<A, B><<<BLOCKS, 512>>> (p, n); }, d, ITERS);            // A B A B
  const double ia = (double) BLOCKS * 8 * ITERS * 8 * 8, ig = 2 * ia, ifn = (double) BLOCKS * 8 * ITERS * 16 * 8;
  const double ca = ta / ia / ref * 2.2, cb = tb / ia / ref * 2.2, cg = tg / ig / ref * 2.2, cf = tf / ifn / ref * 2.2;
  printf ("%-22s %5.2f   %-22s %5.2f   mean %5.2f   groups of 8: %5.2f   alternating: %5.2f   (cycles per wave64 instruction, v_add_u32 = 2.2)\n", na, ca, nb, cb, (ca + cb) / 2, cg, cf);
}
int main ()
{
  u32 *d; (void) hipMalloc (&d, 4096);
  const float tr = timed ([] (u32 *p, int n) { k<6, 6, 1, 0><<<BLOCKS, 512>>> (p, n); }, d, ITERS);
  const double ref = tr / ((double) BLOCKS * 8 * ITERS * 8 * 8);
  pair<0, 1> ("v_mad_i32_i16", "v_perm_b32", d, ref);
  pair<0, 5> ("v_mad_i32_i16", "v_and_b32 literal", d, ref);
  pair<0, 2> ("v_mad_i32_i16", "v_lerp_u8", d, ref);
  pair<0, 3> ("v_mad_i32_i16", "v_sat_pk_u8_i16", d, ref);
  pair<0, 4> ("v_mad_i32_i16", "v_dot4_u32_u8", d, ref);
  pair<0, 13> ("v_mad_i32_i16", "v_lshlrev_b32_sdwa", d, ref);
  pair<1, 3> ("v_perm_b32", "v_sat_pk_u8_i16", d, ref);
  pair<1, 2> ("v_perm_b32", "v_lerp_u8", d, ref);
  pair<1, 5> ("v_perm_b32", "v_and_b32 literal", d, ref);
  pair<2, 14> ("v_lerp_u8", "v_xor_b32 literal", d, ref);
  pair<7, 8> ("v_cvt_f32_ubyte1", "v_fmac_f32", d, ref);
  pair<9, 8> ("v_rndne_f32", "v_fmac_f32", d, ref);
  pair<9, 10> ("v_rndne_f32", "v_pk_mul_f32", d, ref);
  pair<7, 10> ("v_cvt_f32_ubyte1", "v_pk_mul_f32", d, ref);
  pair<11, 10> ("v_cndmask_b32", "v_pk_mul_f32", d, ref);
  pair<12, 10> ("v_cvt_pk_u8_f32", "v_pk_mul_f32", d, ref);
  pair<16, 10> ("v_pk_fma_f32", "v_pk_mul_f32", d, ref);
  pair<15, 8> ("v_exp_f32", "v_fmac_f32", d, ref);
  pair<15, 7> ("v_exp_f32", "v_cvt_f32_ubyte1", d, ref);
  pair<17, 8> ("v_fma_mix_f32", "v_fmac_f32", d, ref);
  pair<8, 5> ("v_fmac_f32", "v_and_b32 literal", d, ref);
  return 0;
}
