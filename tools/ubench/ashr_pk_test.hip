// tools/ubench/ashr_pk_test.hip — semantics of gfx950's v_ashr_pk_u8_i32 (both destination halves) + issue rates of the
// byte-pipeline opcodes the convertscale kernels lean on (diagnostic tool, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_sem (const int *a, const int *b, uint32_t *out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t lo = 0xdeadbeefu, hi = 0xdeadbeefu, both = 0xdeadbeefu;
  asm volatile ("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "+v"(lo) : "v"(a[i]), "v"(b[i]));
  asm volatile ("v_ashr_pk_u8_i32 %0, %1, %2, 16 op_sel:[0,0,0,1]" : "+v"(hi) : "v"(a[i]), "v"(b[i]));
  asm volatile ("v_ashr_pk_u8_i32 %0, %1, %2, 16\n\tv_ashr_pk_u8_i32 %0, %2, %1, 16 op_sel:[0,0,0,1]" : "+v"(both) : "v"(a[i]), "v"(b[i]));
  out[3 * i] = lo; out[3 * i + 1] = hi; out[3 * i + 2] = both;
}
#define REP8(X) X X X X X X X X
#define OPDEF(ID, STR) \
  template <> __device__ __forceinline__ uint32_t op<ID> (uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm volatile (STR : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
template <int OP> __device__ __forceinline__ uint32_t op (uint32_t a, uint32_t b, uint32_t c);
OPDEF (0, "v_add_u32 %0, %1, %2")
OPDEF (1, "v_ashr_pk_u8_i32 %0, %1, %2, 16")
OPDEF (2, "v_perm_b32 %0, %1, %2, %3")
OPDEF (3, "v_lerp_u8 %0, %1, %2, %3")
OPDEF (4, "v_sat_pk_u8_i16 %0, %1")
OPDEF (5, "v_mad_i32_i16 %0, %1, %2, %3")
OPDEF (6, "v_dot4_u32_u8 %0, %1, %2, %3")
OPDEF (7, "v_mul_hi_i32_i24 %0, %1, %2")
OPDEF (8, "v_alignbyte_b32 %0, %1, %2, 2")
OPDEF (9, "v_pack_b32_f16 %0, %1, %2")
OPDEF (10, "v_cvt_pk_u8_f32 %0, %1, 1, %2")
OPDEF (11, "v_bfe_i32 %0, %1, 8, 8")
OPDEF (12, "v_mul_hi_u32 %0, %1, %2")
OPDEF (13, "v_mad_u32_u16 %0, %1, %2, %3")
OPDEF (14, "v_and_b32 %0, 0xff, %1")
OPDEF (15, "v_alignbit_b32 %0, %1, %2, 16")
OPDEF (16, "v_lshl_or_b32 %0, %1, 16, %2")
OPDEF (17, "v_xad_u32 %0, %1, %2, %3")
OPDEF (18, "v_sub_u16 %0, %1, %2")
OPDEF (19, "v_max_u16 %0, %1, %2")
OPDEF (20, "v_ashrrev_i16 %0, 4, %1")
OPDEF (21, "v_pk_mul_lo_u16 %0, %1, %2")
OPDEF (22, "v_pk_mad_u16 %0, %1, %2, %3")
OPDEF (23, "v_fma_f16 %0, %1, %2, %3")
OPDEF (24, "v_add_f16 %0, %1, %2")
OPDEF (25, "v_cvt_f16_u16 %0, %1")
OPDEF (26, "v_cvt_u16_f16 %0, %1")
OPDEF (27, "v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD")
OPDEF (28, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2")
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
static uint32_t *d;
template <int OP> void run (const char *name)
{
  const int iters = 1000, blocks = 256 * 8;
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  k<OP><<<blocks, 256>>> (d, 10);
  (void) hipDeviceSynchronize ();
  (void) hipEventRecord (e0);
  k<OP><<<blocks, 256>>> (d, iters);
  (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
  float ms; (void) hipEventElapsedTime (&ms, e0, e1);
  double ops = (double) blocks * 256 * iters * 64;
  printf ("%-34s %8.3f ms  %8.1f Glane-ops/s\n", name, ms, ops / ms / 1e6);
}
#define RUN(ID, NAME) run<ID> (NAME);
int main ()
{
  const int n = 8;
  const int ha[n] = { 0x00100000, (int) 0xfff00000, 0x01230000, 0x00ff0000, 0x0100ffff, 0x007f8000, (int) 0x80000000, 0x7fffffff };
  const int hb[n] = { 0x00200000, 0x00300000, 0x00450000, (int) 0xffff0000, 0x00010000, 0x00808000, 0x00aa0000, 0x00bb0000 };
  int *da, *db; uint32_t *dout, hout[3 * n];
  (void) hipMalloc (&da, sizeof ha); (void) hipMalloc (&db, sizeof hb); (void) hipMalloc (&dout, sizeof hout);
  (void) hipMemcpy (da, ha, sizeof ha, hipMemcpyHostToDevice); (void) hipMemcpy (db, hb, sizeof hb, hipMemcpyHostToDevice);
  k_sem<<<1, 64>>> (da, db, dout, n);
  (void) hipMemcpy (hout, dout, sizeof hout, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; i++) printf ("a=%08x b=%08x  lo-dest=%08x  hi-dest(op_sel3)=%08x  both(a,b | b,a)=%08x\n", ha[i], hb[i], hout[3 * i], hout[3 * i + 1], hout[3 * i + 2]);
  (void) hipMalloc (&d, 4096);
  RUN (0, "v_add_u32") RUN (1, "v_ashr_pk_u8_i32") RUN (2, "v_perm_b32") RUN (3, "v_lerp_u8") RUN (4, "v_sat_pk_u8_i16") RUN (5, "v_mad_i32_i16") RUN (6, "v_dot4_u32_u8")
  RUN (7, "v_mul_hi_i32_i24") RUN (8, "v_alignbyte_b32") RUN (9, "v_pack_b32_f16") RUN (10, "v_cvt_pk_u8_f32") RUN (11, "v_bfe_i32") RUN (12, "v_mul_hi_u32") RUN (13, "v_mad_u32_u16")
  RUN (14, "v_and_b32 inline") RUN (15, "v_alignbit_b32") RUN (16, "v_lshl_or_b32") RUN (17, "v_xad_u32") RUN (18, "v_sub_u16") RUN (19, "v_max_u16") RUN (20, "v_ashrrev_i16")
  RUN (21, "v_pk_mul_lo_u16") RUN (22, "v_pk_mad_u16") RUN (23, "v_fma_f16") RUN (24, "v_add_f16") RUN (25, "v_cvt_f16_u16") RUN (26, "v_cvt_u16_f16") RUN (27, "v_and_b32_sdwa BYTE_2") RUN (28, "v_mov_b32_sdwa BYTE_2")
  return 0;
}
