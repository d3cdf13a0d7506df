// Shared helpers for the gfx950 kernels of libdmhomo_hip.so.
#pragma once
#include <initializer_list>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dmhomo_hip.h"

void dmh_set_error(const char* fmt, ...);

#define DMH_REQUIRE(cond, ...)   \
  do {                           \
    if (!(cond)) {               \
      dmh_set_error(__VA_ARGS__); \
      return DMH_EINVAL;         \
    }                            \
  } while (0)

#define DMH_CHECK_LAUNCH(name)                                        \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      dmh_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return DMH_ELAUNCH;                                             \
    }                                                                 \
  } while (0)

// size arguments of the pure host entry points (dmh_*_floats, dmh_conv_tiles, ...): a dimension outside (0, 2^20] — a caller's
// uninitialised or overflowed int — is answered with -1 instead of entering the size arithmetic (found by the host-side
// sanitizer build, make asan: a 2^30 channel count overflowed the int64 product)
static inline bool dmh_dims_ok(std::initializer_list<long long> dims, long long lo = 1, long long hi = 1 << 20) {
  for (long long d : dims)
    if (d < lo || d > hi) return false;
  return true;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// x * sigmoid(x) written as torch's CPU SiLU does it: x / (1 + exp(-x))
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// same function on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1-2 ulp): used on the sampling / forward
// path — conv prologues, the residual epilogue and gn_silu_residual — where its VALU cost is visible (+0.8 % images/s)
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- cross-lane reductions without the LDS crossbar.  hipcc compiles every __shfl_xor to ds_bpermute_b32 (an LDS-pipe
// operation with its round trip, one after the other in a reduction); on gfx950 the same butterflies exist as vector-ALU
// operations: DPP controls inside a row of 16 lanes (quad_perm = xor 1 / xor 2; row_half_mirror and row_mirror stand in for
// xor 4 / xor 8 once the quads resp. halves are uniform), v_permlane16_swap for xor 16 and v_permlane32_swap for xor 32.
// Every lane ends with the same bits (the butterfly is symmetric); row16_sum adds xor 1, 2, 4, 8 in that order.
template <int CTRL>
__device__ __forceinline__ unsigned dmh_dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned dmh_xor16(unsigned v, unsigned* other) {  // (own row's value, partner row's value)
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  *other = r[1];
  return r[0];
}
__device__ __forceinline__ unsigned dmh_xor32(unsigned v, unsigned* other) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  *other = r[1];
  return r[0];
}
// sum / max over the 16 lanes of a row (every lane gets the result)
__device__ __forceinline__ float row16_sum(float v) {
  v += __uint_as_float(dmh_dpp<0xB1>(__float_as_uint(v)));   // quad_perm [1,0,3,2]
  v += __uint_as_float(dmh_dpp<0x4E>(__float_as_uint(v)));   // quad_perm [2,3,0,1]
  v += __uint_as_float(dmh_dpp<0x141>(__float_as_uint(v)));  // row_half_mirror
  v += __uint_as_float(dmh_dpp<0x140>(__float_as_uint(v)));  // row_mirror
  return v;
}
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  v = max(v, dmh_dpp<0xB1>(v));
  v = max(v, dmh_dpp<0x4E>(v));
  v = max(v, dmh_dpp<0x141>(v));
  v = max(v, dmh_dpp<0x140>(v));
  return v;
}
// over the four rows of a wave: lanes l, l ^ 16, l ^ 32, l ^ 48 (every lane gets the result)
__device__ __forceinline__ float rows_sum(float v) {
  unsigned o;
  unsigned a = dmh_xor16(__float_as_uint(v), &o);
  v = __uint_as_float(a) + __uint_as_float(o);
  a = dmh_xor32(__float_as_uint(v), &o);
  return __uint_as_float(a) + __uint_as_float(o);
}
__device__ __forceinline__ float rows_max(float v) {
  unsigned o;
  unsigned a = dmh_xor16(__float_as_uint(v), &o);
  v = fmaxf(__uint_as_float(a), __uint_as_float(o));
  a = dmh_xor32(__float_as_uint(v), &o);
  return fmaxf(__uint_as_float(a), __uint_as_float(o));
}
__device__ __forceinline__ unsigned rows_max_u32(unsigned v) {
  unsigned o;
  unsigned a = dmh_xor16(v, &o);
  v = max(a, o);
  a = dmh_xor32(v, &o);
  return max(a, o);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return rows_max_u32(row16_max_u32(v)); }
// sum over aligned groups of LPP = 2 .. 64 lanes (a power of two)
template <int LPP>
__device__ __forceinline__ float lanes_sum(float v) {
  static_assert(LPP >= 1 && LPP <= 64 && (LPP & (LPP - 1)) == 0, "a power of two up to 64 lanes");
  if (LPP >= 2) v += __uint_as_float(dmh_dpp<0xB1>(__float_as_uint(v)));
  if (LPP >= 4) v += __uint_as_float(dmh_dpp<0x4E>(__float_as_uint(v)));
  if (LPP >= 8) v += __uint_as_float(dmh_dpp<0x141>(__float_as_uint(v)));
  if (LPP >= 16) v += __uint_as_float(dmh_dpp<0x140>(__float_as_uint(v)));
  unsigned o;
  if (LPP >= 32) {
    const unsigned a = dmh_xor16(__float_as_uint(v), &o);
    v = __uint_as_float(a) + __uint_as_float(o);
  }
  if (LPP >= 64) {
    const unsigned a = dmh_xor32(__float_as_uint(v), &o);
    v = __uint_as_float(a) + __uint_as_float(o);
  }
  return v;
}

// ---- the two fp16 pieces of x * s:  h = fp16(x * s),  r = fp16(x * s - h)  (pairs packed into dwords, low half first).
// r is formed from the UNROUNDED product (one fma), so h + r carries 22 bits of x * s whatever s is (hipcc's own sequence
// for the C expression has been seen to form the two pieces from differently rounded products: linattn_fused.hip).
// Round 3: six instructions per pair that cost 21.6 SIMD cycles instead of four v_fma_mixlo/hi_f16 that cost 32.7 — on
// gfx950 the fp16-writing mix instructions issue at a QUARTER of the fp32 rate (8.2 cycles per wave64 instruction, like a
// transcendental), v_cvt_pk_f16_f32 and v_fma_mix_f32 at half rate, v_mul_f32 at full rate (tools/micro/valu_rate.hip,
// profiles/r03_micro_valu_rate.txt).  For a power-of-two s (every block scale) the pieces are bit for bit the old ones
// (tools/micro/split_equiv.hip); for an arbitrary s (the softmax normaliser of linattn_qo_kernel) h is now rounded
// fp32 -> fp16 from the rounded product, and r — still taken from the exact product — absorbs the difference.
__device__ __forceinline__ void dmh_split2(float x0, float x1, float s, unsigned& h, unsigned& r) {
  float t0, t1, u0, u1;
  asm("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(x0), "v"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(x1), "v"(s));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(t0), "v"(t1));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(u0) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(u1) : "v"(x1), "v"(s), "v"(h));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(u0), "v"(u1));
}
typedef _Float16 dmh_half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void dmh_split8(const float (&x)[8], float s, dmh_half8& h, dmh_half8& r) {
  uint4 hh, rr;
  dmh_split2(x[0], x[1], s, hh.x, rr.x);
  dmh_split2(x[2], x[3], s, hh.y, rr.y);
  dmh_split2(x[4], x[5], s, hh.z, rr.z);
  dmh_split2(x[6], x[7], s, hh.w, rr.w);
  // The pieces are MFMA operands, and hipcc's hazard recognizer does not see registers written inside inline asm: without
  // wait states here an MFMA scheduled right behind the split read them before they were written (seen as NaN / a lost
  // second piece in one 16-pixel block of linattn_qo_kernel, moving with the schedule).
  asm volatile("s_nop 3" : "+v"(hh.x), "+v"(hh.y), "+v"(hh.z), "+v"(hh.w), "+v"(rr.x), "+v"(rr.y), "+v"(rr.z), "+v"(rr.w));
  h = __builtin_bit_cast(dmh_half8, hh);
  r = __builtin_bit_cast(dmh_half8, rr);
}

// ---- row subsets of a batched launch (de-duplication of the classifier-free-guidance passes, CFG:403-425).
// rows == nullptr: logical row j of the launch is physical row (sample slot) j, j < B.  Otherwise rows[0] = n <= B active
// rows and rows[1 + j] = the physical row of logical row j < n (dmh_rows_from_keep).  The launch keeps its full B-row grid —
// n is device data, so the launch can sit in a captured graph — and a workgroup whose logical row is >= n retires before
// it touches memory; tensors are indexed by PHYSICAL row, so nothing is gathered or scattered.
__device__ __forceinline__ int dmh_rows_n(const int32_t* rows, int B) { return rows ? rows[0] : B; }
__device__ __forceinline__ int dmh_rows_phys(const int32_t* rows, int j) { return rows ? rows[1 + j] : j; }

// input-channel chunk width of the conv variants (must agree between pack and kernel)
static inline int conv_kc(int KH, int stride) { return (KH == 7 || stride == 2) ? 16 : 32; }
