// tools/ubench/valu_occ.hip — does the VALU reach its issue rate with 4 waves per SIMD (what 68 KB of LDS per 512-lane workgroup leaves the video
// filter's sharpening kernel) and with 3 dependent chains per wave (its blur sums)?  Diagnostic tool, not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CHAINS, int MIX> __global__ __launch_bounds__ (512) void k (float *out, int iters, float w)
{
  extern __shared__ float lds[];
  float r[CHAINS];
  double rr[CHAINS];
  for (int i = 0; i < CHAINS; i++) { r[i] = threadIdx.x * 0.001f + i; rr[i] = r[i]; }
  const float x = blockIdx.x * 0.5f + 1.0f, w2 = w * 0.25f;
  const double xx = x;
  unsigned long long ww = ((unsigned long long) __float_as_uint (w) << 32) | __float_as_uint (w);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int i = 0; i < CHAINS; i++) {
        if (MIX == 0) asm volatile ("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(x), "v"(w));
        else if (MIX == 1) { if ((u & 3) == 3) asm volatile ("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[i])); else asm volatile ("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(x), "v"(w)); }
        else if (MIX == 2) asm volatile ("v_fmac_f32 %0, %2, %1" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 3) asm volatile ("v_fma_f32 %0, %2, %1, %0" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 4) asm volatile ("v_mul_f32 %0, %2, %0" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 5) asm volatile ("v_fma_f32 %0, %1, %1, %0" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 6) asm volatile ("v_fma_f32 %0, %0, %2, %3" : "+v"(r[i]) : "v"(x), "s"(w), "s"(w2));
        else if (MIX == 7) asm volatile ("v_add_f32 %0, %2, %0" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 8) asm volatile ("v_fmac_f32 %0, 0.5, %1" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 10) asm volatile ("v_mad_i32_i16 %0, %0, %2, %1" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 11) asm volatile ("v_mad_i32_i16 %0, %0, %1, %1" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 12) asm volatile ("v_and_b32 %0, %2, %0" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 13) asm volatile ("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 14) asm volatile ("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 15) asm volatile ("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(x), "s"(w));
        else if (MIX == 9) asm volatile ("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(rr[i]) : "v"(xx), "s"(ww));
      }
  }
  float s = 0;
  for (int i = 0; i < CHAINS; i++) s += r[i] + (float) rr[i];
  if (s == 12345.f) out[0] = s + lds[threadIdx.x];
}
template <int CHAINS, int MIX> void run (const char *name, size_t lds_bytes, float *d)
{
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  (void) hipFuncSetAttribute ((const void *) k<CHAINS, MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<CHAINS, MIX><<<blocks, 512, lds_bytes>>> (d, 10, 0.5f);
  (void) hipDeviceSynchronize ();
  float best = 1e9f;
  for (int t = 0; t < 3; t++) {
    (void) hipEventRecord (e0);
    k<CHAINS, MIX><<<blocks, 512, lds_bytes>>> (d, iters, 0.5f);
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double instr = (double) blocks * 8 * iters * 16 * CHAINS;      // wave-instructions
  printf ("%-58s lds %6zu KB/WG  %8.3f ms  %6.2f ns*SIMD per wave-instruction (x clock = cycles): at 2.3 GHz %.2f cycles  (%s)\n", name, lds_bytes / 1024, best,
          best * 1e6 * 1024 / instr, best * 1e6 * 1024 / instr * 2.3, hipGetErrorString (hipGetLastError ()));
}
int main ()
{
  float *d; (void) hipMalloc (&d, 4096);
  const size_t occ8 = 1024, occ4 = 70 * 1024, occ2 = 150 * 1024;          // 512-lane workgroups: 4 / 2 / 1 per CU -> 8 / 4 / 2 waves per SIMD
  run<8, 0> ("v_fmac x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 0> ("v_fmac x 8 chains, 4 waves/SIMD", occ4, d);
  run<8, 0> ("v_fmac x 8 chains, 2 waves/SIMD", occ2, d);
  run<3, 0> ("v_fmac x 3 chains, 8 waves/SIMD", occ8, d);
  run<3, 0> ("v_fmac x 3 chains, 4 waves/SIMD", occ4, d);
  run<3, 0> ("v_fmac x 3 chains, 2 waves/SIMD", occ2, d);
  run<3, 2> ("v_fmac_f32 v, SGPR, v  x 3 chains, 4 waves/SIMD", occ4, d);
  run<8, 2> ("v_fmac_f32 v, SGPR, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 3> ("v_fma_f32 v, SGPR, v, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 4> ("v_mul_f32 v, SGPR, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 7> ("v_add_f32 v, SGPR, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 5> ("v_fma_f32 v, v, v, v (VOP3, no sgpr) x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 6> ("v_fma_f32 v, v, SGPR, SGPR' x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 8> ("v_fmac_f32 v, 0.5 (inline const), v x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 10> ("v_mad_i32_i16 v, v, SGPR, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 11> ("v_mad_i32_i16 v, v, v, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 12> ("v_and_b32 v, SGPR, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 13> ("v_and_b32 v, literal, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 14> ("v_perm_b32 v, v, v, SGPR  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 15> ("v_perm_b32 v, v, v, v  x 8 chains, 8 waves/SIMD", occ8, d);
  run<8, 9> ("v_pk_fma_f32 v[2], v[2], SGPR pair (one instr = 2 fma) x 8, 8 waves/SIMD", occ8, d);
  run<3, 1> ("3 v_fmac : 1 v_cvt_f32_ubyte x 3 chains, 8 waves/SIMD", occ8, d);
  run<3, 1> ("3 v_fmac : 1 v_cvt_f32_ubyte x 3 chains, 4 waves/SIMD", occ4, d);
  run<1, 0> ("v_fmac x 1 chain, 8 waves/SIMD", occ8, d);
  run<1, 0> ("v_fmac x 1 chain, 4 waves/SIMD", occ4, d);
  return 0;
}
