// micro-benchmark (round 3): the issue model a gfx950 SIMD presents to ONE and to TWO resident waves — what a wave's
// instruction costs in cycles when it is independent / dependent, vector / transcendental / MFMA / cross-lane, alone or beside
// a partner wave running the same stream.  Cycles by s_memtime inside the kernel (shader clock), per wave, averaged.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/issue_model.hip -o /tmp/issue_model && /tmp/issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

enum { FMA_IND, FMA_DEP, EXP_IND, MFMA_IND, MFMA_DEP, MIX1, MIX2, MIX4, MIX8, SWAP_DEP, MIXLO_DEP, LDS_DEP, FMA_IND2, PKFMA_IND, LDS128_IND, CVT_IND, MX_E1, MX_E1F2, MX_E1F4, MX_H_E1, MX_H_E1F2, MX_MIX1, MX_MIX2, EXPFMA, NTEST };
static const char* NAME[NTEST] = {"v_fma_f32 independent (8 chains)", "v_fma_f32 dependent (1 chain)", "v_exp_f32 independent",
                                  "mfma 16x16x32 f16 independent (8 acc)", "mfma 16x16x32 f16 dependent (1 acc)",
                                  "1 mfma + 1 v_fma", "1 mfma + 2 v_fma", "1 mfma + 4 v_fma", "1 mfma + 8 v_fma",
                                  "v_permlane32_swap dependent", "v_fma_mixlo_f16 dependent", "ds_read_b32 dependent (pointer chase)",
                                  "v_fma_f32 independent (2 chains)", "v_pk_fma_f32 independent (8 chains)",
                                  "ds_read_b128 independent (8 in flight)", "v_cvt_pkrtz_f16_f32 independent",
                                  "1 mfma + 1 v_exp", "1 mfma + 1 v_exp + 2 v_fma", "1 mfma + 1 v_exp + 4 v_fma",
                                  "2 mfma + 1 v_exp", "2 mfma + 1 v_exp + 2 v_fma", "1 mfma + 1 v_fma_mixlo", "1 mfma + 2 v_fma_mixlo",
                                  "1 v_exp + 2 v_fma (no mfma)"};
// instructions per unrolled body (for the per-instruction figure)
static const int PER[NTEST] = {64, 64, 64, 32, 32, 32 * 2, 32 * 3, 32 * 5, 32 * 9, 32, 64, 32, 64, 64, 64, 64, 32 * 2, 32 * 4, 32 * 6, 16 * 3, 16 * 5, 32 * 2, 32 * 3, 32 * 3};

template <int T>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
  __shared__ unsigned lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = ((i * 37 + 11) & 1023) * 4;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(lane * 0.02f - j); }
  float4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
  unsigned p = lane * 4, h = 0;
  typedef float float2v __attribute__((ext_vector_type(2)));
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  float2v pk[8], pc = {1.0001f, 0.5f};
  for (int i = 0; i < 8; ++i) pk[i] = float2v{lane * 0.01f, (float)i};
  unsigned hh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint4v q[8];
  for (int i = 0; i < 8; ++i) q[i] = uint4v{0, 0, 0, 0};
  const float c = 1.0001f, d = 0.5f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (T == FMA_IND) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
    } else if (T == FMA_IND2) {
#pragma unroll
      for (int r = 0; r < 32; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
    } else if (T == FMA_DEP) {
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(c), "v"(d));
    } else if (T == EXP_IND) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if (T == MFMA_IND) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    } else if (T == MFMA_DEP) {
#pragma unroll
      for (int r = 0; r < 32; ++r) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[0], 0, 0, 0);
    } else if (T == MIX1 || T == MIX2 || T == MIX4 || T == MIX8) {
      constexpr int NV = T == MIX1 ? 1 : T == MIX2 ? 2 : T == MIX4 ? 4 : 8;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        acc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[r & 7], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
      }
    } else if (T >= MX_E1 && T <= EXPFMA) {
      constexpr int NM = (T == MX_H_E1 || T == MX_H_E1F2) ? 2 : (T == EXPFMA ? 0 : 1);
      constexpr int NE = (T == MX_MIX1 || T == MX_MIX2) ? 0 : 1;
      constexpr int NF = (T == MX_E1F2 || T == MX_H_E1F2 || T == EXPFMA) ? 2 : (T == MX_E1F4 ? 4 : 0);
      constexpr int NX = T == MX_MIX1 ? 1 : (T == MX_MIX2 ? 2 : 0);
#pragma unroll
      for (int r = 0; r < (NM == 2 ? 16 : 32); ++r) {
#pragma unroll
        for (int m = 0; m < NM; ++m)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(r * NM + m) & 7]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r & 3]));
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4 + ((r * NF + i) & 3)]) : "v"(c), "v"(d));
#pragma unroll
        for (int i = 0; i < NX; ++i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(hh[(r * NX + i) & 7]) : "v"(v[1]), "v"(c));
      }
    } else if (T == SWAP_DEP) {
#pragma unroll
      for (int r = 0; r < 32; ++r) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(p), "+v"(h));
    } else if (T == MIXLO_DEP) {
#pragma unroll
      for (int r = 0; r < 64; ++r) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(h) : "v"(v[1]), "v"(c));
    } else if (T == PKFMA_IND) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i]) : "v"(pc));
    } else if (T == CVT_IND) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hh[i]) : "v"(v[i]), "v"(c));
    } else if (T == LDS128_IND) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(lane * 16), "n"(i * 1024 % 3072) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else if (T == LDS_DEP) {
#pragma unroll
      for (int r = 0; r < 32; ++r) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(p)::"memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0] + acc[i][3] + pk[i][0] + pk[i][1] + (float)hh[i] + (float)(q[i][0] ^ q[i][3]);
  if (s == 123.456f || p == 0xffffffffu || h == 0x12345u) out[0] = 1;   // keep everything alive
  if (lane == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (lane == 0) out[1 + 256 * 16 + blockIdx.x * 16 + (threadIdx.x >> 6)] = r1 - r0;
}

template <int T>
void run(unsigned long long* dout, int waves_per_simd) {
  const int iters = 2000, threads = 256 * waves_per_simd, blocks = 256;
  hipMemset(dout, 0, (1 + 2 * blocks * 16) * 8);
  k<T><<<blocks, threads>>>(dout, iters);
  k<T><<<blocks, threads>>>(dout, iters);
  hipDeviceSynchronize();
  static unsigned long long hbuf[1 + 2 * 256 * 16];
  hipMemcpy(hbuf, dout, sizeof(hbuf), hipMemcpyDeviceToHost);
  // a workgroup's waves do not progress evenly (the arbiter favours the older wave of a SIMD): the SIMD's throughput is
  // waves / SIMD x instructions over the LAST wave's time, so print the slowest and the fastest wave of a workgroup too
  double sum = 0, rsum = 0, mx = 0, mn = 0;
  int nw = 0;
  for (int b = 0; b < blocks; ++b) {
    double bmx = 0, bmn = 1e30;
    for (int w = 0; w < threads / 64; ++w) {
      const double t = (double)hbuf[1 + b * 16 + w];
      sum += t; rsum += (double)hbuf[1 + 256 * 16 + b * 16 + w]; ++nw;
      bmx = t > bmx ? t : bmx; bmn = t < bmn ? t : bmn;
    }
    mx += bmx; mn += bmn;
  }
  const double tick = sum / nw / iters, ns = rsum / nw / iters * 10.0;   // s_memrealtime counts at 100 MHz
  const double slow = mx / blocks / iters / PER[T], fast = mn / blocks / iters / PER[T];
  printf("%-42s %d wave/SIMD: %6.2f cycles/instr/wave (fastest %6.2f slowest %6.2f) => SIMD issues one per %5.2f cycles  [%.2f GHz]\n",
         NAME[T], waves_per_simd, tick / PER[T], fast, slow, slow / waves_per_simd, tick / ns);
}

int main() {
  unsigned long long* dout;
  hipMalloc(&dout, (1 + 2 * 256 * 16) * 8);
#define BOTH(T) run<T>(dout, 1); run<T>(dout, 2); run<T>(dout, 4);
  BOTH(FMA_IND) BOTH(FMA_IND2) BOTH(FMA_DEP) BOTH(EXP_IND) BOTH(MFMA_IND) BOTH(MFMA_DEP) BOTH(MIX1) BOTH(MIX2) BOTH(MIX4) BOTH(MIX8)
  BOTH(SWAP_DEP) BOTH(MIXLO_DEP) BOTH(LDS_DEP) BOTH(PKFMA_IND) BOTH(CVT_IND) BOTH(LDS128_IND)
  BOTH(MX_E1) BOTH(MX_E1F2) BOTH(MX_E1F4) BOTH(MX_H_E1) BOTH(MX_H_E1F2) BOTH(MX_MIX1) BOTH(MX_MIX2) BOTH(EXPFMA)
  return 0;
}
