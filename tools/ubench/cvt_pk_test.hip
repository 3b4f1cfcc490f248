// tools/ubench/cvt_pk_test.hip — does v_cvt_pk_u8_f32 round to nearest even and saturate like
// (uint) rintf (clamp (x, 0, 255))?  (diagnostic: decides whether quant8 can use it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k (const float *x, unsigned *a, unsigned *b, int n)
{
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  a[i] = __builtin_amdgcn_cvt_pk_u8_f32 (x[i], 0u, 0u);
  b[i] = (unsigned) __float2int_rn (fminf (fmaxf (x[i], 0.0f), 255.0f));
}
int main ()
{
  std::vector<float> h;
  for (int k = -3; k < 260; k++) for (float d : { -0.5f, -0.49999f, -0.25f, 0.0f, 0.25f, 0.49999f, 0.5f, 0.50001f }) h.push_back (k + d);
  for (int i = 0; i < 200000; i++) h.push_back ((float) (rand () % 300000) / 1000.0f - 20.0f);
  for (int v = 0; v < 256; v++) for (int w = 0; w < 256; w += 5) h.push_back (((float) v * (1.0f / 255.0f) + (float) w * (1.0f / 255.0f)) * 0.5f * 255.0f);
  int n = h.size ();
  float *dx; unsigned *da, *db;
  (void) hipMalloc (&dx, n * 4); (void) hipMalloc (&da, n * 4); (void) hipMalloc (&db, n * 4);
  (void) hipMemcpy (dx, h.data (), n * 4, hipMemcpyHostToDevice);
  k<<<(n + 255) / 256, 256>>> (dx, da, db, n);
  std::vector<unsigned> a (n), b (n);
  (void) hipMemcpy (a.data (), da, n * 4, hipMemcpyDeviceToHost); (void) hipMemcpy (b.data (), db, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) if (a[i] != b[i]) { if (bad < 10) printf ("x=%.6f cvt_pk=%u rne=%u\n", h[i], a[i], b[i]); bad++; }
  printf ("n=%d mismatches=%d\n", n, bad);
  return 0;
}
