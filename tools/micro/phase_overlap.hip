// micro-benchmark (round 3): two waves per SIMD, each alternating a MATRIX phase (LM back-to-back MFMAs) with a VECTOR phase
// (LV instructions of one kind) — the shape of the conv / LinearAttention kernels (stage, barrier, multiply).  The partner wave
// starts either in phase or half a period out of phase.  Printed: cycles per period per wave, against the two bounds
// (perfect overlap of one wave's vector phase with the other's matrix phase; none).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/phase_overlap.hip -o tools/micro/phase_overlap && tools/micro/phase_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

enum { V_FMA, V_EXP, V_MIX, NV };
static const char* VN[NV] = {"v_fma_f32", "v_exp_f32", "v_fma_mixlo_f16"};

template <int V, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned long long* out, int periods, int lm, int lv, int dephase) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(lane * 0.02f - j); }
  float4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
  unsigned hh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float c = 1.0001f, d = 0.5f;
  unsigned long long t0, t1;
  auto vec = [&](int n) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (V == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
        if (V == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (V == V_MIX) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(hh[i]) : "v"(v[i]), "v"(c));
      }
    }
  };
  __syncthreads();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (dephase && wave >= 4) vec(lv);          // the partner wave starts with its vector phase
  for (int pd = 0; pd < periods; ++pd) {
    for (int it = 0; it < lm; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    vec(lv);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0] + acc[i][3] + (float)hh[i];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) out[1 + blockIdx.x * 16 + wave] = t1 - t0;
}

template <int V, int WAVES>
double once(unsigned long long* dout, int lm, int lv, int dephase) {
  const int periods = 200, blocks = 256;
  static unsigned long long h[1 + 256 * 16];
  for (int rep = 0; rep < 2; ++rep) k<V, WAVES><<<blocks, WAVES * 64>>>(dout, periods, lm, lv, dephase);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int b = 0; b < blocks; ++b) {
    double bm = 0;
    for (int w = 0; w < WAVES; ++w) bm = (double)h[1 + b * 16 + w] > bm ? (double)h[1 + b * 16 + w] : bm;
    mx += bm;
  }
  return mx / blocks / periods;
}

template <int V>
void run(unsigned long long* dout, int lm, int lv) {
  const double one = once<V, 4>(dout, lm, lv, 0), one_m = once<V, 4>(dout, lm, 0, 0), one_v = once<V, 4>(dout, 0, lv, 0);
  const double two = once<V, 8>(dout, lm, lv, 0), two_d = once<V, 8>(dout, lm, lv, 1);
  const double four = once<V, 16>(dout, lm, lv, 0);
  printf("%-16s %3d mfma + %3d vector | 1 wave/SIMD: %7.0f (matrix %6.0f + vector %6.0f) | 2 waves/SIMD per 2 periods: in phase %7.0f  out of phase %7.0f"
         "  [overlap bound %6.0f, serial %6.0f] | 4 waves per 4 periods: %7.0f [bound %6.0f]\n",
         VN[V], lm * 8, lv * 8, one, one_m, one_v, two, two_d, 2 * (one_m > one_v ? one_m : one_v), 2 * one, four,
         4 * (one_m > one_v ? one_m : one_v));
}

int main() {
  unsigned long long* dout;
  (void)hipMalloc(&dout, (1 + 256 * 16) * 8);
  run<V_FMA>(dout, 16, 32);    // 128 mfma (2048 cycles) | 256 fma
  run<V_FMA>(dout, 16, 64);
  run<V_FMA>(dout, 2, 8);      // short phases
  run<V_FMA>(dout, 1, 2);
  run<V_EXP>(dout, 16, 16);
  run<V_EXP>(dout, 16, 32);
  run<V_MIX>(dout, 16, 16);
  run<V_MIX>(dout, 16, 32);
  return 0;
}
