// micro-benchmark (round 3): does the order in which a wave's MFMAs visit its accumulators change what the chip can sustain
// under its power cap?  Every SIMD of the chip runs two waves of v_mfma_f32_16x16x32_f16 on RANDOM fp16 operands (8 A and
// 8 B fragments in registers, 8 accumulators), for milliseconds, in two orders:
//   round-robin   acc0, acc1, ..., acc7, acc0, ...        (every accumulator read from / written to the register file)
//   chained       acc0 x3, acc1 x3, ...                   (dependent MFMAs back to back: the pipe forwards the accumulator)
// Same MFMA count, same operands.  Printed: wall time, in-kernel cycles, effective clock.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_chain_power.hip -o tools/micro/mfma_chain_power && tools/micro/mfma_chain_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

// MODE: 0 round-robin; 1 chains of 3; 2 chains of 6; 3 chains of 12; 4 one chain of 24 on a single accumulator per group;
//       5 chains of 3 with the SAME A fragment inside a chain; 6 chains of 3 with the same A and the same B
template <int MODE>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ ops, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  half8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 64 + lane) * 2]; b[i] = ops[(i * 64 + lane) * 2 + 1]; }
  float4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 7]), "v"(b[(i + 2 * r) & 7]));
    } else if (MODE <= 4) {
      constexpr int L = MODE == 1 ? 3 : MODE == 2 ? 6 : MODE == 3 ? 12 : 24;
#pragma unroll
      for (int i = 0; i < 24 / L; ++i)
#pragma unroll
        for (int r = 0; r < L; ++r)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 7]), "v"(b[(i + 2 * r) & 7]));
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[MODE == 6 ? i : (i + 2 * r) & 7]));
    }
    if ((it & 63) == 63)   // keep the sums finite
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] *= 1e-3f;
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// the same FLOPs on v_mfma_f32_32x32x16_f16 (half the operand bytes, twice the accumulator bytes per multiply-add): 4 accumulators
// of 16 registers, round-robin or chains of 3
typedef float float16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k32(const half8* __restrict__ ops, float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  half8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 64 + lane) * 2]; b[i] = ops[(i * 64 + lane) * 2 + 1]; }
  float16v acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 7]), "v"(b[(i + 2 * r) & 7]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r)
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 7]), "v"(b[(i + 2 * r) & 7]));
    }
    if ((it & 63) == 63)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] *= 1e-3f;
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  const int blocks = 256, iters = 20000;
  half8* ops;
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&ops, 8 * 64 * 2 * sizeof(half8));
  (void)hipMalloc(&out, blocks * 512 * 4);
  (void)hipMalloc(&cyc, blocks * 8 * 8);
  _Float16* h = (_Float16*)malloc(8 * 64 * 2 * 8 * 2);
  srand(3);
  for (int i = 0; i < 8 * 64 * 2 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 100.f);
  (void)hipMemcpy(ops, h, 8 * 64 * 2 * 8 * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  static unsigned long long hc[256 * 8];
  static const char* MN[7] = {"round-robin", "chains of 3", "chains of 6", "chains of 12", "chains of 24", "3, same A", "3, same A and B"};
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 7; ++mode) {
      (void)hipEventRecord(e0);
      switch (mode) {
        case 0: k<0><<<blocks, 512>>>(ops, out, cyc, iters); break;
        case 1: k<1><<<blocks, 512>>>(ops, out, cyc, iters); break;
        case 2: k<2><<<blocks, 512>>>(ops, out, cyc, iters); break;
        case 3: k<3><<<blocks, 512>>>(ops, out, cyc, iters); break;
        case 4: k<4><<<blocks, 512>>>(ops, out, cyc, iters); break;
        case 5: k<5><<<blocks, 512>>>(ops, out, cyc, iters); break;
        default: k<6><<<blocks, 512>>>(ops, out, cyc, iters); break;
      }
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
      double c = 0;
      for (int i = 0; i < blocks * 8; ++i) c += (double)hc[i];
      c /= blocks * 8;
      const double mf = (double)blocks * 8 * iters * 24;
      printf("%-16s %8.3f ms  %10.0f cycles per wave (%.2f per MFMA and SIMD)  effective clock %.3f GHz  %.0f TFLOP/s\n",
             MN[mode], ms, c, c / (iters * 24.0) / 2.0, c / (ms * 1e6), mf * 16384 / (ms * 1e-3) / 1e12);
    }
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      (void)hipEventRecord(e0);
      if (mode == 0) k32<0><<<blocks, 512>>>(ops, out, cyc, iters);   // 12 MFMAs of twice the FLOPs per iteration: the same work
      else k32<1><<<blocks, 512>>>(ops, out, cyc, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double mf = (double)blocks * 8 * iters * 12;
      printf("32x32x16 %-16s %8.3f ms  %.0f TFLOP/s\n", mode ? "chains of 3" : "round-robin", ms, mf * 32768 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
