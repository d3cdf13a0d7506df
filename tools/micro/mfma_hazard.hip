// How many wait states does a vector instruction need after v_mfma_f32_16x16x32_f16 before it may read the result?
// (hipcc puts `s_nop 4` between such an MFMA and the first VALU reader.)  The consumer here is a v_mul_f32 placed behind
// N wait states written by hand; the MFMA's operands are made "fresh" by preceding dependent VALU work, and a second
// independent MFMA in front keeps the pipe busy, as in linattn_qo_kernel.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_hazard tools/micro/mfma_hazard.hip && /tmp/mfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int N>
__global__ void k(const half8* a, const half8* b, const float4v* c, float4v* ref, float4v* got) {
  const int l = threadIdx.x;
  half8 av = a[l], bv = b[l];
  float4v cv = c[l], c2 = c[l];
  ref[l] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, cv, 0, 0, 0) * 2.0f;
  float4v d = cv, e = c2;
  float r0;
  asm volatile(
      "v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n\t"   // an independent MFMA in front (pipe busy)
      "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"
      "s_nop %4"
      : "+v"(d), "+v"(e)
      : "v"(av), "v"(bv), "n"(N));
  asm volatile("v_mul_f32 %0, 2.0, %1" : "=v"(r0) : "v"(d[0]));
  asm volatile("s_nop 7\n\ts_nop 7" ::);
  float4v o = d * 2.0f;
  o[0] = r0;
  got[l] = o;
}

template <int N>
int run(half8* da, half8* db, float4v* dc, float4v* dr, float4v* dg) {
  float4v r[64], g[64];
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, da, db, dc, dr, dg);
  (void)hipMemcpy(r, dr, sizeof r, hipMemcpyDeviceToHost);
  (void)hipMemcpy(g, dg, sizeof g, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += r[i][0] != g[i][0];
  printf("s_nop %d before the reader: %d of 64 lanes read a wrong first element (ref %g got %g)\n", N, bad, r[3][0], g[3][0]);
  return bad;
}

int main() {
  half8 ha[64], hb[64];
  float4v hc[64];
  srand(1);
  for (int i = 0; i < 64; ++i) {
    for (int j = 0; j < 8; ++j) {
      ha[i][j] = (_Float16)((rand() % 2001 - 1000) / 100.0f);
      hb[i][j] = (_Float16)((rand() % 2001 - 1000) / 100.0f);
    }
    for (int j = 0; j < 4; ++j) hc[i][j] = (rand() % 2001 - 1000) / 10.0f;
  }
  half8 *da, *db;
  float4v *dc, *dr, *dg;
  (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dc, sizeof hc);
  (void)hipMalloc(&dr, sizeof hc); (void)hipMalloc(&dg, sizeof hc);
  (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  (void)hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
  run<0>(da, db, dc, dr, dg); run<1>(da, db, dc, dr, dg); run<2>(da, db, dc, dr, dg); run<3>(da, db, dc, dr, dg);
  run<4>(da, db, dc, dr, dg); run<5>(da, db, dc, dr, dg); run<6>(da, db, dc, dr, dg); run<7>(da, db, dc, dr, dg);
  return 0;
}
