// micro-benchmark (round 3): does the OTHER wave of a SIMD make progress while one wave keeps the matrix pipe busy?
// A 512-thread workgroup per CU: waves 0-3 (one per SIMD) run back-to-back MFMAs, waves 4-7 (their partners) run a second stream
// of N instructions of one kind.  Each wave times itself (s_memtime); printed: both streams alone, both together.
// Perfect overlap: together = max(alone); a shared issue port / pipe: together = sum.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cross_wave.hip -o tools/micro/cross_wave && tools/micro/cross_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

enum { P_FMA, P_EXP, P_MIX, P_CVT, P_LDSR, P_LDSW, P_GLD, P_SWAP, P_MFMA, NP };
static const char* PN[NP] = {"v_fma_f32 x64", "v_exp_f32 x64", "v_fma_mixlo_f16 x64", "v_cvt_pkrtz x64", "ds_read_b128 x64",
                             "ds_write_b64 x64", "global_load_dwordx4 x32 (L2 hits)", "v_permlane32_swap x64", "mfma 16x16x32 x32"};

template <int P>
__global__ __launch_bounds__(512) void k(unsigned long long* out, const float* g, int iters, int run_a, int run_b) {
  __shared__ unsigned lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool is_a = wave < 4;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(lane * 0.01f + j); b[j] = (_Float16)(lane * 0.02f - j); }
  float4v acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
  unsigned hh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint4v q[8];
  for (int i = 0; i < 8; ++i) q[i] = uint4v{0, 0, 0, 0};
  const float c = 1.0001f, d = 0.5f;
  const float* gp = g + (size_t)blockIdx.x * 4096 + (wave & 3) * 1024 + lane * 4;
  unsigned long long t0 = 0, t1 = 0;
  if ((is_a && run_a) || (!is_a && run_b)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
      if (is_a || P == P_MFMA) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
      } else if (P == P_FMA) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
      } else if (P == P_EXP) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      } else if (P == P_MIX) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(hh[i]) : "v"(v[i]), "v"(c));
      } else if (P == P_CVT) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hh[i]) : "v"(v[i]), "v"(c));
      } else if (P == P_LDSR) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(lane * 16), "n"(i * 1024) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      } else if (P == P_LDSW) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(lane * 8 + (wave & 3) * 8192), "v"(*(unsigned long long*)&q[i]), "n"(i * 512) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      } else if (P == P_GLD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(q[i]) : "v"(gp), "n"(i * 256 % 4096) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else if (P == P_SWAP) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(hh[i]), "+v"(q[i][0]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + acc[i][0] + acc[i][3] + (float)hh[i] + (float)(q[i][0] ^ q[i][3]);
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) out[1 + blockIdx.x * 8 + wave] = t1 - t0;
}

template <int P>
void run(unsigned long long* dout, const float* g) {
  const int iters = 1000, blocks = 256;
  static unsigned long long h[1 + 256 * 8];
  double res[3][2];
  for (int mode = 0; mode < 3; ++mode) {   // 0: matrix waves alone, 1: partner waves alone, 2: together
    const int ra = mode != 1, rb = mode != 0;
    for (int rep = 0; rep < 2; ++rep) k<P><<<blocks, 512>>>(dout, g, iters, ra, rb);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[1 + b * 8 + w];
    res[mode][0] = sa / (blocks * 4) / iters;
    res[mode][1] = sb / (blocks * 4) / iters;
  }
  printf("%-36s matrix wave (32 mfma): alone %7.1f  together %7.1f | partner: alone %7.1f  together %7.1f   [max %.0f, sum %.0f]\n", PN[P],
         res[0][0], res[2][0], res[1][1], res[2][1], res[0][0] > res[1][1] ? res[0][0] : res[1][1], res[0][0] + res[1][1]);
}

int main() {
  unsigned long long* dout;
  float* g;
  (void)hipMalloc(&dout, (1 + 256 * 8) * 8);
  (void)hipMalloc(&g, 256 * 4096 * 4 + 65536);
  (void)hipMemset(g, 0, 256 * 4096 * 4 + 65536);
  printf("cycles per loop body, per wave; 256 workgroups x 512 threads (2 waves per SIMD)\n");
  run<P_FMA>(dout, g); run<P_EXP>(dout, g); run<P_MIX>(dout, g); run<P_CVT>(dout, g); run<P_LDSR>(dout, g); run<P_LDSW>(dout, g);
  run<P_GLD>(dout, g); run<P_SWAP>(dout, g); run<P_MFMA>(dout, g);
  return 0;
}
