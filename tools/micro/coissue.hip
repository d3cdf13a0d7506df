// micro-benchmark: can ONE wave per SIMD hide VALU work in the shadow of its own MFMA stream (hipcc-scheduled)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/coissue.hip -o tools/micro/coissue && tools/micro/coissue
// modes: 0 = MFMA only, 1 = VALU only, 2 = both in one loop body (same scheduling region); MF = 16 (16x16x32) or 32 (32x32x16)
// per iteration: 24 (MF32) / 48 (MF16) MFMAs = 768 matrix-pipe cycles, and NV independent "SiLU + fp16 split" element chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE, int MF, int NV>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)(in[threadIdx.x % 64 + j] * 0.01f);
    b[j] = (_Float16)(in[threadIdx.x % 64 + j + 8] * 0.01f);
  }
  floatx16 acc32[4] = {};
  float4v acc16[8] = {};
  float v[NV], s = 0.f;
  for (int j = 0; j < NV; ++j) v[j] = in[threadIdx.x + j * 256];
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) {
      constexpr int NM = MF == 32 ? 24 : 48;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (MF == 32) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[i & 3], 0, 0, 0);
        else acc16[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc16[i & 7], 0, 0, 0);
#pragma unroll
        for (int j = i * NV / NM; j < (i + 1) * NV / NM; ++j) {
          float t = fmaf(1.01f, v[j], 0.001f);
          float y = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
          float xs = y * 1024.f;
          _Float16 h1 = (_Float16)xs;
          float rr = (xs - (float)h1) * 2048.f;
          _Float16 h2 = (_Float16)rr;
          s += (float)h2;
          v[j] = y + 0.5f;
        }
        if (MODE == 4) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 14 * NV / NM, 0);         // the VALU of this slot
        }
      }
      continue;
    }
    if (MODE != 1) {
      if (MF == 32) {
#pragma unroll
        for (int i = 0; i < 24; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[i & 3], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 48; ++i) acc16[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc16[i & 7], 0, 0, 0);
      }
    }
    if (MODE != 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {               // ~12 VALU per element: fma, silu (exp, rcp), scale, split
        float t = fmaf(1.01f, v[j], 0.001f);
        float y = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t));
        float xs = y * 1024.f;
        _Float16 h1 = (_Float16)xs;
        float r = (xs - (float)h1) * 2048.f;
        _Float16 h2 = (_Float16)r;
        s += (float)h2;
        v[j] = y + 0.5f;
      }
    }
  }
  // MODE 3: source-level interleave, one MFMA then the VALU chain of NV/24 elements; MODE 4: the same + sched_group_barrier
  float r = s;
  for (int i = 0; i < 4; ++i) r += acc32[i][0];
  for (int i = 0; i < 8; ++i) r += acc16[i][0];
  for (int j = 0; j < NV; ++j) r += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int MF, int NV>
float run(float* out, float* in, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE, MF, NV><<<256, 256>>>(out, in, 10);
  hipEventRecord(e0);
  k<MODE, MF, NV><<<256, 256>>>(out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&in, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 7919) % 1000) / 500.f - 1.f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int it = 2000;
  printf("one workgroup of 4 waves per CU (1 wave per SIMD), %d iterations; us\n", it);
#define ROW(MF, NV)                                                                                           \
  printf("MF=%d NV=%3d : mfma only %8.1f   valu only %8.1f   both %8.1f  interleaved %8.1f  + sched_group %8.1f\n", MF, NV, \
         run<0, MF, NV>(out, in, it), run<1, MF, NV>(out, in, it), run<2, MF, NV>(out, in, it), run<3, MF, NV>(out, in, it), \
         run<4, MF, NV>(out, in, it));
  ROW(32, 24)
  ROW(32, 48)
  ROW(16, 48)
  ROW(16, 96)
  return 0;
}
