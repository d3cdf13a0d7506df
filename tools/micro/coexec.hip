// Microbenchmark: do fp32 MFMAs issued by one wave overlap VALU / LDS work issued by the
// other wave resident on the same SIMD?  One 512-thread workgroup per CU (waves 0-3 and 4-7
// share SIMDs 0-3).  Roles: A = waves 0-3, B = waves 4-7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

enum { NONE = 0, MFMA16 = 1, VALU = 2, LDSR = 3, MFMA32 = 4, PKVALU = 5, BF32 = 6, BF32DEP = 7, BF16 = 8, SFMA = 9, IOPS = 10 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ void work(int iters, float* out, float* lds) {
  const int lane = threadIdx.x & 63;
  if (ROLE == MFMA16) {
    f4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = lane * 0.001f, b = lane * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x] = s;
  } else if (ROLE == MFMA32) {
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = lane * 0.001f, b = lane * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    out[threadIdx.x] = s;
  } else if (ROLE == BF32 || ROLE == BF32DEP) {
    // 8 x v_mfma_f32_32x32x16_bf16 per iteration; BF32: 4 accumulators round robin, BF32DEP: one accumulator
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(lane * 0.001f + j); b[j] = (__bf16)(lane * 0.002f - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = ROLE == BF32 ? (i & 3) : 0;
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
      }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    out[threadIdx.x] = s;
  } else if (ROLE == BF16) {
    f4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(lane * 0.001f + j); b[j] = (__bf16)(lane * 0.002f - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
  } else if (ROLE == SFMA) {
    // 64 scalar (unpacked) v_fma_f32 per iteration, forced through inline asm so the SLP vectoriser cannot pack them
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
    const float c = 1.0001f, d = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
  } else if (ROLE == IOPS) {
    // 64 integer/bit ops per iteration (v_and_b32 / v_sub_f32-free): the bf16 split's mask + perm mix
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lane * 77u + i;
    const unsigned m = 0xffff0000u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
          asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(m));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
  } else if (ROLE == VALU) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)  // 64 v_fma per iteration (== 8 MFMA16 = 256 cycles of MFMA)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
  } else if (ROLE == PKVALU) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{lane * 0.01f + i, lane * 0.02f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)  // 32 v_pk_add per iteration
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] + f2{0.5f, 0.25f};
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[threadIdx.x] = s;
  } else if (ROLE == LDSR) {
    f4 s4 = f4{0, 0, 0, 0};
    const f4* l4 = reinterpret_cast<const f4*>(lds);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // 8 ds_read_b128 per iteration
        f4 t = l4[((it * 8 + r) * 64 + lane) & 1023];
        s4 += t;
      }
    }
    out[threadIdx.x] = s4[0] + s4[1] + s4[2] + s4[3];
  }
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(int iters, float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  float* o = out + blockIdx.x * 512;
  if (threadIdx.x < 256) work<RA>(iters, o, lds);
  else work<RB>(iters, o, lds);
}

template <int RA, int RB>
float run(const char* name, int iters, float* out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<RA, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<RA, RB><<<256, 512, 100 * 1024>>>(iters, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<RA, RB><<<256, 512, 100 * 1024>>>(iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.1f us\n", name, ms / 5 * 1e3);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int it = 20000;
  run<BF32, NONE>("bf16 32x32x16 x8/it alone", it, out);
  run<BF32DEP, NONE>("bf16 32x32x16 dependent", it, out);
  run<BF16, NONE>("bf16 16x16x32 x8/it alone", it, out);
  run<BF32, BF32>("bf16 32x32x16 + same", it, out);
  run<BF32, VALU>("bf16 32x32x16 + valu", it, out);
  run<BF32, LDSR>("bf16 32x32x16 + lds", it, out);
  run<BF32DEP, BF32DEP>("bf16 dep + dep", it, out);
  run<SFMA, NONE>("scalar v_fma x64/it alone", it, out);
  run<IOPS, NONE>("int and/perm x64/it alone", it, out);
  run<BF32, SFMA>("bf16 32x32x16 + scalar fma", it, out);
  run<BF32, IOPS>("bf16 32x32x16 + int ops", it, out);
  run<BF16, SFMA>("bf16 16x16x32 + scalar fma", it, out);
  run<BF16, BF16>("bf16 16x16x32 + same", it, out);
  run<MFMA16, SFMA>("f32 mfma16 + scalar fma", it, out);
  run<MFMA16, IOPS>("f32 mfma16 + int ops", it, out);
  run<MFMA16, NONE>("mfma16x16x4 alone", it, out);
  run<MFMA32, NONE>("mfma32x32x2 alone", it, out);
  run<VALU, NONE>("valu(64 fma/it) alone", it, out);
  run<PKVALU, NONE>("pk_add(32/it) alone", it, out);
  run<LDSR, NONE>("lds(8 b128/it) alone", it, out);
  run<MFMA16, MFMA16>("mfma16 + mfma16", it, out);
  run<MFMA16, VALU>("mfma16 + valu", it, out);
  run<MFMA32, VALU>("mfma32 + valu", it, out);
  run<MFMA16, PKVALU>("mfma16 + pk_add", it, out);
  run<MFMA16, LDSR>("mfma16 + lds", it, out);
  run<MFMA32, LDSR>("mfma32 + lds", it, out);
  run<VALU, VALU>("valu + valu", it, out);
  run<VALU, LDSR>("valu + lds", it, out);
  return 0;
}
