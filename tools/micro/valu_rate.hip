// micro-benchmark (round 3): SIMD throughput of single vector instructions on gfx950 — 4 waves per SIMD, 8 independent
// destination registers per wave, inline asm; cycles by s_memtime.  Which conversions / mixed-precision fmas are full rate
// (one wave64 instruction per ~2.3 cycles), half rate (~4.3) or quarter rate (~8.2): the fp16 split of the f16x3 kernels is
// made of them.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define OPS(X)                                                                                       \
  X(0, "v_fma_f32", "v_fma_f32 %0, %1, %2, %0")                                                        \
  X(1, "v_mul_f32", "v_mul_f32 %0, %1, %2")                                                            \
  X(2, "v_pk_mul_f32", "v_pk_mul_f32 %3, %4, %4")                                                      \
  X(3, "v_pk_add_f32 (neg)", "v_pk_add_f32 %3, %4, %4 neg_lo:[0,1] neg_hi:[0,1]")                      \
  X(4, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %1")                                                        \
  X(5, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %1")                                                        \
  X(6, "v_cvt_f32_f16 sdwa WORD_1", "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") \
  X(7, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %1, %2")                                        \
  X(8, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %1, %2")                                              \
  X(9, "v_fma_mix_f32 (f16 lo src)", "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")                 \
  X(10, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]")                         \
  X(11, "v_fma_mixhi_f16", "v_fma_mixhi_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]")                         \
  X(12, "v_fma_mixlo_f16 (f32 srcs, 0)", "v_fma_mixlo_f16 %0, %1, %2, 0")                              \
  X(13, "v_pk_mul_f16", "v_pk_mul_f16 %0, %1, %2")                                                     \
  X(14, "v_pk_fma_f16", "v_pk_fma_f16 %0, %1, %2, %0")                                                 \
  X(15, "v_exp_f32", "v_exp_f32 %0, %1")                                                               \
  X(16, "v_rcp_f32", "v_rcp_f32 %0, %1")                                                               \
  X(17, "v_max_f32", "v_max_f32 %0, %1, %2")                                                           \
  X(18, "v_and_b32", "v_and_b32 %0, %1, %2")                                                           \
  X(19, "v_perm_b32", "v_perm_b32 %0, %1, %2, %0")                                                     \
  X(20, "v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(21, "v_permlane32_swap", "v_permlane32_swap_b32 %0, %1")                                           \
  X(22, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %1, %2")                                           \
  X(23, "v_sub_f32 sdwa (f16 hi as f32? no: plain)", "v_sub_f32 %0, %1, %2")
#define NOPS 24

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
  const int lane = threadIdx.x & 63;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float d[8], a = lane * 0.01f + 1.f, b = 1.0001f;
  float2v pd[8], pa = {a, b};
  for (int i = 0; i < 8; ++i) { d[i] = (float)i; pd[i] = float2v{(float)i, 1.f}; }
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(id, name, text) if (OP == id) asm volatile(text : "+v"(d[i]) : "v"(a), "v"(b), "v"(pd[i]), "v"(pa));
        OPS(X)
#undef X
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += d[i] + pd[i][0] + pd[i][1];
  if (s == 123.456f) out[0] = 1;
  if (lane == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(unsigned long long* dout, const char* name) {
  const int iters = 1000, blocks = 256;
  static unsigned long long h[1 + 256 * 16];
  double res[2];
  for (int wi = 0; wi < 2; ++wi) {
    const int waves = wi == 0 ? 4 : 16;
    for (int rep = 0; rep < 2; ++rep) k<OP><<<blocks, waves * 64>>>(dout, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int b = 0; b < blocks; ++b) {
      double bm = 0;
      for (int w = 0; w < waves; ++w) bm = (double)h[1 + b * 16 + w] > bm ? (double)h[1 + b * 16 + w] : bm;
      mx += bm;
    }
    res[wi] = mx / blocks / iters / 64 / (waves / 4);
  }
  printf("%-44s one wave: %5.2f cycles each | 4 waves per SIMD: one per %5.2f cycles\n", name, res[0], res[1]);
}

int main() {
  unsigned long long* dout;
  (void)hipMalloc(&dout, (1 + 256 * 16) * 8);
#define X(id, name, text) run<id>(dout, name);
  OPS(X)
#undef X
  return 0;
}
