// check (round 3): the six-instruction fp16 split (v_mul, v_cvt_pk_f16_f32, v_fma_mix_f32 with the fp16 piece as addend) gives
// bit for bit the pieces of the four-v_fma_mix form it replaces — over random fp32 inputs of every exponent, power-of-two and
// arbitrary multipliers.  Prints the number of differing dwords (0 expected; +0 / -0 counted separately).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/split_equiv.hip -o tools/micro/split_equiv && tools/micro/split_equiv
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ void split_old(float x0, float x1, float s, unsigned& h, unsigned& r) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ void split_new(float x0, float x1, float s, unsigned& h, unsigned& r) {
  float t0, t1;
  asm("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x0), "v"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x1), "v"(s));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(t0), "v"(t1));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(t0) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(t1) : "v"(x1), "v"(s), "v"(h));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(t0), "v"(t1));
}
__global__ void k(const float* x, float s, int n, unsigned long long* diff) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned h0, r0, h1, r1;
  split_old(x[2 * i], x[2 * i + 1], s, h0, r0);
  split_new(x[2 * i], x[2 * i + 1], s, h1, r1);
  if (h0 != h1 || r0 != r1) {
    // sign-of-zero differences: a half equal up to the sign bit with both magnitudes zero
    auto zero_only = [](unsigned a, unsigned b) {
      for (int hf = 0; hf < 2; ++hf) {
        const unsigned ha = (a >> (16 * hf)) & 0xffff, hb = (b >> (16 * hf)) & 0xffff;
        if (ha != hb && !(((ha | hb) & 0x7fff) == 0)) return false;
      }
      return true;
    };
    atomicAdd(&diff[zero_only(h0, h1) && zero_only(r0, r1) ? 1 : 0], 1ull);
    if (!(zero_only(h0, h1) && zero_only(r0, r1)) && atomicAdd(&diff[2], 1ull) < 4)
      printf("x = %a %a  s = %a: old %08x %08x  new %08x %08x\n", x[2 * i], x[2 * i + 1], s, h0, r0, h1, r1);
  }
}
int main() {
  const int n = 1 << 24;
  float* hx = (float*)malloc(n * 4);
  srand(7);
  for (int i = 0; i < n; ++i) {
    unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    if ((b & 0x7f800000u) == 0x7f800000u) b &= 0xbfffffffu;   // no inf / nan
    if (i % 64 == 0) b &= 0x80000000u;                        // some zeros
    if (i % 64 == 1) b &= 0x807fffffu;                        // some fp32 subnormals
    memcpy(&hx[i], &b, 4);
  }
  float* dx;
  unsigned long long* dd;
  (void)hipMalloc(&dx, n * 4);
  (void)hipMalloc(&dd, 24);
  (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  const float scales[] = {1.f, 1024.f, 0x1p-100f, 0x1p100f, 0x1p-20f, 0.7321f, 3.1e-12f, 5.5e11f, 1.0000001f};
  // the product kernels scale so that |x * s| < 2^15; also restrict x to that window for the realistic cases
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      for (int i = 0; i < n; ++i) hx[i] = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, rand() % 40 - 25);
      (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    }
    for (float s : scales) {
      if (pass == 1 && (s < 1e-6f || s > 1e6f)) continue;
      (void)hipMemset(dd, 0, 24);
      k<<<n / 2 / 256, 256>>>(dx, s, n, dd);
      unsigned long long h[3];
      (void)hipMemcpy(h, dd, 24, hipMemcpyDeviceToHost);
      printf("%s inputs, s = %-12g: %llu differing pairs (+ %llu that differ in the sign of a zero only)\n",
             pass ? "windowed" : "all-exponent", s, h[0], h[1]);
    }
  }
  return 0;
}
