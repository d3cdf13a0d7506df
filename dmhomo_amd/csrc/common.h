// Shared helpers for the gfx950 kernels of libdmhomo_hip.so.
#pragma once
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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// x * sigmoid(x) written as torch's CPU SiLU does it: x / (1 + exp(-x))
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// same function on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1-2 ulp): used on the sampling / forward
// path — conv prologues, the residual epilogue and gn_silu_residual — where its VALU cost is visible (+0.8 % images/s)
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// input-channel chunk width of the conv variants (must agree between pack and kernel)
static inline int conv_kc(int KH, int stride) { return (KH == 7 || stride == 2) ? 16 : 32; }
