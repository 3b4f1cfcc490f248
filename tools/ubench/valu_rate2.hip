// tools/ubench/valu_rate2.hip — which gfx950 VALU opcodes issue at the 2-cycle (wave64) rate and which at 4?
// (diagnostic tool, not product code).  Each op is an asm string with %0 dst, %1 %2 %3 VGPR sources.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
#define OPDEF(ID, STR) \
  template <> __device__ __forceinline__ uint32_t op<ID> (uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm volatile (STR : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
template <int OP> __device__ __forceinline__ uint32_t op (uint32_t a, uint32_t b, uint32_t c);
OPDEF (0, "v_add_u32 %0, %1, %2")
OPDEF (1, "v_sub_u32 %0, %1, %2")
OPDEF (2, "v_or_b32 %0, %1, %2")
OPDEF (3, "v_xor_b32 %0, %1, %2")
OPDEF (4, "v_lshlrev_b32 %0, 3, %1")
OPDEF (5, "v_lshrrev_b32 %0, 8, %1")
OPDEF (6, "v_ashrrev_i32 %0, 16, %1")
OPDEF (7, "v_mul_u32_u24 %0, %1, %2")
OPDEF (8, "v_mul_i32_i24 %0, %1, %2")
OPDEF (9, "v_min_i32 %0, %1, %2")
OPDEF (10, "v_max_i32 %0, %1, %2")
OPDEF (11, "v_cndmask_b32 %0, %1, %2, vcc")
OPDEF (12, "v_add_f32 %0, %1, %2")
OPDEF (13, "v_mul_f32 %0, %1, %2")
OPDEF (14, "v_fmac_f32 %0, %1, %2")
OPDEF (15, "v_mov_b32 %0, %1")
OPDEF (16, "v_cvt_f32_ubyte1 %0, %1")
OPDEF (17, "v_cvt_u32_f32 %0, %1")
OPDEF (18, "v_cvt_f32_i32 %0, %1")
OPDEF (19, "v_floor_f32 %0, %1")
OPDEF (20, "v_fma_f32 %0, %1, %2, %3")
OPDEF (21, "v_mad_u32_u24 %0, %1, %2, %3")
OPDEF (22, "v_add3_u32 %0, %1, %2, %3")
OPDEF (23, "v_and_or_b32 %0, %1, %2, %3")
OPDEF (24, "v_bfe_u32 %0, %1, 8, 8")
OPDEF (25, "v_bfi_b32 %0, %1, %2, %3")
OPDEF (26, "v_lshl_add_u32 %0, %1, 2, %2")
OPDEF (27, "v_or3_b32 %0, %1, %2, %3")
OPDEF (28, "v_mul_lo_u32 %0, %1, %2")
OPDEF (29, "v_pk_add_u16 %0, %1, %2")
OPDEF (30, "v_pk_add_i16 %0, %1, %2 clamp")
OPDEF (32, "v_pk_max_i16 %0, %1, %2")
OPDEF (33, "v_pk_lshrrev_b16 %0, 8, %1")
OPDEF (34, "v_dot2_i32_i16 %0, %1, %2, %3")
OPDEF (35, "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
OPDEF (36, "v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:WORD_0")
OPDEF (37, "v_add_u32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf")
OPDEF (38, "v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf")
OPDEF (39, "v_and_b32 %0, 0xffff0000, %1")
OPDEF (40, "v_add_u16 %0, %1, %2")
OPDEF (41, "v_mad_u16 %0, %1, %2, %3")
OPDEF (42, "v_mul_lo_u16 %0, %1, %2")
OPDEF (43, "v_add_co_u32 %0, vcc, %1, %2")
OPDEF (45, "v_mul_f16 %0, %1, %2")
OPDEF (46, "v_pk_fma_f16 %0, %1, %2, %3")
OPDEF (47, "v_cvt_pkrtz_f16_f32 %0, %1, %2")
OPDEF (48, "v_max_f32 %0, %1, %2")
OPDEF (49, "v_med3_f32 %0, %1, %2, %3")
OPDEF (50, "v_rndne_f32 %0, %1")
OPDEF (51, "v_sad_u8 %0, %1, %2, %3")
OPDEF (52, "v_msad_u8 %0, %1, %2, %3")
OPDEF (53, "v_mad_i32_i24 %0, %1, %2, 64")
OPDEF (54, "v_add_f32 %0, %1, %2 clamp")
OPDEF (55, "v_fma_f32 %0, %1, %2, %3 clamp")
OPDEF (56, "v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
OPDEF (57, "v_subrev_u32 %0, %1, %2")
OPDEF (58, "v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3")
OPDEF (59, "v_lshlrev_b16 %0, 3, %1")
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
  (void) hipMalloc (&d, 4096);
  RUN (0, "v_add_u32") RUN (1, "v_sub_u32") RUN (57, "v_subrev_u32") RUN (2, "v_or_b32") RUN (3, "v_xor_b32") RUN (39, "v_and_b32 literal") RUN (4, "v_lshlrev_b32") RUN (5, "v_lshrrev_b32")
  RUN (6, "v_ashrrev_i32") RUN (7, "v_mul_u32_u24") RUN (8, "v_mul_i32_i24") RUN (9, "v_min_i32") RUN (10, "v_max_i32") RUN (11, "v_cndmask_b32")
  RUN (12, "v_add_f32") RUN (54, "v_add_f32 clamp(VOP3)") RUN (13, "v_mul_f32") RUN (48, "v_max_f32") RUN (14, "v_fmac_f32") RUN (15, "v_mov_b32") RUN (16, "v_cvt_f32_ubyte1") RUN (58, "v_cvt_f32_ubyte0_sdwa") RUN (17, "v_cvt_u32_f32")
  RUN (18, "v_cvt_f32_i32") RUN (19, "v_floor_f32") RUN (50, "v_rndne_f32") RUN (20, "v_fma_f32") RUN (55, "v_fma_f32 clamp") RUN (49, "v_med3_f32") RUN (21, "v_mad_u32_u24") RUN (53, "v_mad_i32_i24 literal") RUN (22, "v_add3_u32") RUN (23, "v_and_or_b32")
  RUN (24, "v_bfe_u32") RUN (25, "v_bfi_b32") RUN (26, "v_lshl_add_u32") RUN (27, "v_or3_b32") RUN (28, "v_mul_lo_u32") RUN (29, "v_pk_add_u16")
  RUN (30, "v_pk_add_i16 clamp") RUN (32, "v_pk_max_i16") RUN (33, "v_pk_lshrrev_b16") RUN (34, "v_dot2_i32_i16")
  RUN (35, "v_add_u32_sdwa") RUN (36, "v_mul_u32_u24_sdwa") RUN (56, "v_mul_f32_sdwa") RUN (37, "v_add_u32_dpp") RUN (38, "v_mov_b32_dpp")
  RUN (40, "v_add_u16") RUN (59, "v_lshlrev_b16") RUN (41, "v_mad_u16") RUN (42, "v_mul_lo_u16") RUN (43, "v_add_co_u32") RUN (45, "v_mul_f16") RUN (46, "v_pk_fma_f16") RUN (47, "v_cvt_pkrtz_f16_f32")
  RUN (51, "v_sad_u8") RUN (52, "v_msad_u8")
  return 0;
}
