// tools/ubench/valu_rate3.hip — issue rates of the float opcodes the `metal`-numerics kernels lean on (round 3): hardware transcendentals,
// packed-f32 VOP3P forms (what hipcc's SLP vectoriser emits), v_fma_mix_f32 (f16 operands into an f32 fma), fract / min / max / cndmask / compares.
// Diagnostic tool, not product code.  Same harness as valu_rate2.hip: 8 independent chains per lane, 2048 blocks x 256 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
typedef uint32_t u32;
typedef unsigned long long u64;
template <int OP> __device__ __forceinline__ u32 op (u32 a, u32 b, u32 c);
#define OPDEF(ID, STR) \
  template <> __device__ __forceinline__ u32 op<ID> (u32 a, u32 b, u32 c) { u32 d; asm volatile (STR : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
OPDEF (0, "v_add_f32 %0, %1, %2")
OPDEF (1, "v_fma_f32 %0, %1, %2, %3")
OPDEF (2, "v_exp_f32 %0, %1")
OPDEF (3, "v_log_f32 %0, %1")
OPDEF (4, "v_rcp_f32 %0, %1")
OPDEF (5, "v_rsq_f32 %0, %1")
OPDEF (6, "v_sqrt_f32 %0, %1")
OPDEF (7, "v_fract_f32 %0, %1")
OPDEF (8, "v_min_f32 %0, %1, %2")
OPDEF (9, "v_max3_f32 %0, %1, %2, %3")
OPDEF (10, "v_cndmask_b32 %0, %1, %2, s[10:11]")
OPDEF (11, "v_cmp_lt_f32 vcc, %1, %2\n v_mov_b32 %0, %1")
OPDEF (12, "v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")
OPDEF (13, "v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1]")
OPDEF (14, "v_cvt_f32_f16 %0, %1")
OPDEF (15, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
OPDEF (16, "v_cvt_i32_f32 %0, %1")
OPDEF (17, "v_mul_f32 %0, %1, %2 clamp")
OPDEF (18, "v_mul_f32 %0, 0x40490fdb, %1")
OPDEF (19, "v_fmaak_f32 %0, %1, %2, 0x40490fdb")
OPDEF (20, "v_sub_f32 %0, 1.0, %1")
OPDEF (21, "v_mul_legacy_f32 %0, %1, %2")
OPDEF (22, "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
OPDEF (23, "v_ldexp_f32 %0, %1, %2")
OPDEF (24, "v_frexp_mant_f32 %0, %1")
OPDEF (25, "v_cvt_pk_u8_f32 %0, %1, 1, %2")
OPDEF (26, "v_cvt_f32_ubyte2 %0, %1")
OPDEF (27, "v_add_f32 %0, |%1|, %2")
OPDEF (28, "v_floor_f32 %0, %1")
OPDEF (29, "v_trunc_f32 %0, %1")
OPDEF (30, "v_mad_u32_u24 %0, %1, %2, %3")
OPDEF (31, "v_max_f32 %0, %1, %2")
// 64-bit (register pair) forms: packed f32
template <int OP> __device__ __forceinline__ u64 op2 (u64 a, u64 b, u64 c);
#define OP2DEF(ID, STR) \
  template <> __device__ __forceinline__ u64 op2<ID> (u64 a, u64 b, u64 c) { u64 d; asm volatile (STR : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
OP2DEF (0, "v_pk_fma_f32 %0, %1, %2, %3")
OP2DEF (1, "v_pk_mul_f32 %0, %1, %2")
OP2DEF (2, "v_pk_add_f32 %0, %1, %2")
OP2DEF (3, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")
OP2DEF (4, "v_pk_mov_b32 %0, %1, %2")
template <int OP> __global__ __launch_bounds__ (256) void k (u32 *out, int iters)
{
  u32 r[8];
  for (int i = 0; i < 8; i++) r[i] = 0x3f000000u + threadIdx.x * 7 + i;
  u32 b = 0x3f800000u | blockIdx.x, c = 0x3e000000u | threadIdx.x;
  asm volatile ("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11");
  for (int it = 0; it < iters; it++) {
    REP8 (r[0] = op<OP> (r[0], b, c); r[1] = op<OP> (r[1], b, c); r[2] = op<OP> (r[2], b, c); r[3] = op<OP> (r[3], b, c);
          r[4] = op<OP> (r[4], b, c); r[5] = op<OP> (r[5], b, c); r[6] = op<OP> (r[6], b, c); r[7] = op<OP> (r[7], b, c);)
  }
  u32 s = 0;
  for (int i = 0; i < 8; i++) s ^= r[i];
  if (s == 0x12345) out[0] = s;
}
template <int OP> __global__ __launch_bounds__ (256) void k2 (u32 *out, int iters)
{
  u64 r[8];
  for (int i = 0; i < 8; i++) r[i] = 0x3f0000003f000000ull + threadIdx.x * 7 + i;
  u64 b = 0x3f8000003f800000ull | blockIdx.x, c = 0x3e0000003e000000ull | threadIdx.x;
  for (int it = 0; it < iters; it++) {
    REP8 (r[0] = op2<OP> (r[0], b, c); r[1] = op2<OP> (r[1], b, c); r[2] = op2<OP> (r[2], b, c); r[3] = op2<OP> (r[3], b, c);
          r[4] = op2<OP> (r[4], b, c); r[5] = op2<OP> (r[5], b, c); r[6] = op2<OP> (r[6], b, c); r[7] = op2<OP> (r[7], b, c);)
  }
  u64 s = 0;
  for (int i = 0; i < 8; i++) s ^= r[i];
  if (s == 0x12345) out[0] = (u32) s;
}
static u32 *d;
static double base_ms = 0;
template <typename K> void timeit (K kern, const char *name, int lanes_per_op)
{
  const int iters = 1000, blocks = 256 * 8;
  hipEvent_t e0, e1; (void) hipEventCreate (&e0); (void) hipEventCreate (&e1);
  kern<<<blocks, 256>>> (d, 10);
  (void) hipDeviceSynchronize ();
  float best = 1e9f;
  for (int t = 0; t < 3; t++) {
    (void) hipEventRecord (e0);
    kern<<<blocks, 256>>> (d, iters);
    (void) hipEventRecord (e1); (void) hipEventSynchronize (e1);
    float ms; (void) hipEventElapsedTime (&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  if (base_ms == 0) base_ms = best;
  double ops = (double) blocks * 256 * iters * 64 * lanes_per_op;
  printf ("%-44s %8.3f ms  %8.1f Glane-ops/s   %.2f cycles per wave64 instruction (v_add_f32 = 2)\n", name, best, ops / best / 1e6, 2.0 * best / base_ms);
}
#define RUN(ID, NAME) timeit (k<ID>, NAME, 1);
#define RUN2(ID, NAME) timeit (k2<ID>, NAME, 2);
int main ()
{
  (void) hipMalloc (&d, 4096);
  RUN (0, "v_add_f32") RUN (1, "v_fma_f32") RUN (17, "v_mul_f32 clamp") RUN (18, "v_mul_f32 literal") RUN (19, "v_fmaak_f32 (literal)") RUN (20, "v_sub_f32 1.0, x") RUN (27, "v_add_f32 |x|")
  RUN (21, "v_mul_legacy_f32")
  RUN (2, "v_exp_f32") RUN (3, "v_log_f32") RUN (4, "v_rcp_f32") RUN (5, "v_rsq_f32") RUN (6, "v_sqrt_f32")
  RUN (7, "v_fract_f32") RUN (28, "v_floor_f32") RUN (29, "v_trunc_f32") RUN (8, "v_min_f32") RUN (31, "v_max_f32") RUN (9, "v_max3_f32") RUN (10, "v_cndmask_b32 (sgpr mask)") RUN (11, "v_cmp_lt_f32 + v_mov")
  RUN (12, "v_fma_mix_f32 (f16 lo, f32, f16 lo)") RUN (13, "v_fma_mix_f32 (f16 hi, f32, f16 lo)") RUN (14, "v_cvt_f32_f16") RUN (15, "v_cvt_f32_f16_sdwa WORD_1")
  RUN (16, "v_cvt_i32_f32") RUN (22, "v_cvt_f32_u32_sdwa WORD_1") RUN (23, "v_ldexp_f32") RUN (24, "v_frexp_mant_f32") RUN (25, "v_cvt_pk_u8_f32") RUN (26, "v_cvt_f32_ubyte2") RUN (30, "v_mad_u32_u24")
  RUN2 (0, "v_pk_fma_f32 (2 lanes per op)") RUN2 (1, "v_pk_mul_f32") RUN2 (2, "v_pk_add_f32") RUN2 (3, "v_pk_fma_f32 op_sel_hi (scalar broadcast)") RUN2 (4, "v_pk_mov_b32")
  return 0;
}
