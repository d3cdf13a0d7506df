// Does v_mfma_f32_16x16x32_f16 tolerate a destination that is the same registers as its B (or A) operand?
// hipcc allocates such overlaps (linattn_fused.hip, conv_f16x3.hip); this runs both forms on random data and compares.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_overlap tools/micro/mfma_overlap.hip && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void k(const half8* a, const half8* b, const float4v* c, float4v* ref, float4v* ovb, float4v* ova, int reps) {
  const int l = threadIdx.x;
  half8 av = a[l], bv = b[l];
  float4v cv = c[l];
  ref[l] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, cv, 0, 0, 0);
  float4v xb = __builtin_bit_cast(float4v, bv);   // destination == B
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %0, %2\n\ts_nop 7\n\ts_nop 7" : "+v"(xb) : "v"(av), "v"(cv));
  ovb[l] = xb;
  float4v xa = __builtin_bit_cast(float4v, av);   // destination == A
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %0, %1, %2\n\ts_nop 7\n\ts_nop 7" : "+v"(xa) : "v"(bv), "v"(cv));
  ova[l] = xa;
}

int main() {
  half8 ha[64], hb[64];
  float4v hc[64], r[64], ob[64], oa[64];
  srand(1);
  for (int i = 0; i < 64; ++i) {
    for (int j = 0; j < 8; ++j) {
      ha[i][j] = (_Float16)((rand() % 2001 - 1000) / 100.0f);
      hb[i][j] = (_Float16)((rand() % 2001 - 1000) / 100.0f);
    }
    for (int j = 0; j < 4; ++j) hc[i][j] = (rand() % 2001 - 1000) / 10.0f;
  }
  half8 *da, *db;
  float4v *dc, *dr, *dob, *doa;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc);
  hipMalloc(&dr, sizeof r); hipMalloc(&dob, sizeof r); hipMalloc(&doa, sizeof r);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dr, dob, doa, 1);
  hipMemcpy(r, dr, sizeof r, hipMemcpyDeviceToHost); hipMemcpy(ob, dob, sizeof r, hipMemcpyDeviceToHost);
  hipMemcpy(oa, doa, sizeof r, hipMemcpyDeviceToHost);
  int badb = 0, bada = 0;
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < 4; ++j) {
      badb += r[i][j] != ob[i][j];
      bada += r[i][j] != oa[i][j];
    }
  printf("destination == B: %d of 256 values differ from the non-overlapping MFMA; destination == A: %d differ\n", badb, bada);
  printf("sample ref %g  dst==B %g  dst==A %g\n", r[5][1], ob[5][1], oa[5][1]);
  return 0;
}
