// K6b — noise of the sampling loop keyed by GLOBAL sample index ("results invariant to the number of GPUs").
// The reference draws torch.randn(shape) / randn_like(img) / zeros(B).uniform_() from the process's CUDA stream
// generator (CFG:679, 705, 90): row b of a draw depends on how many rows the process holds.  Here value
// (seed, sample id, draw, element) is a pure function — Philox4x32-10 (Salmon et al., SC'11; Random123's known-answer
// vectors pin it in tests/) with key = seed and counter = (element / 4, draw, sample id lo, sample id hi) — so the
// rows of a shard are the rows the single-GPU run computes for the same samples.  The draw index lives in device
// memory and is advanced by the launch itself (the last workgroup to finish), so a captured denoise step replays
// with fresh noise and no host bookkeeping.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// (0, 1]-open-at-zero uniform of a 32-bit word, as cuRAND / torch's CUDA generator place it: x * 2^-32 + 2^-33
__device__ __forceinline__ float u01(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

// Box-Muller: two words -> two standard normals
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float r = sqrtf(-2.0f * logf(u01(a)));
  float s, c;
  sincospif(2.0f * u01(b), &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// grid: (blocks per sample, B).  kind 0: N(0,1); 1: uniform; 2: the raw words (bit patterns stored in the floats)
__global__ __launch_bounds__(256) void rng_indexed_kernel(float* __restrict__ out, int64_t per, const int64_t* __restrict__ ids,
                                                          unsigned long long* state, int kind) {
  const unsigned long long seed = state[0], draw = state[1];
  const int b = blockIdx.y;
  const unsigned long long sid = (unsigned long long)ids[b];
  float* o = out + (size_t)b * per;
  const int64_t nq = (per + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
    const U4 r = philox4x32_10(U4{(uint32_t)q, (uint32_t)draw, (uint32_t)sid, (uint32_t)(sid >> 32)}, (uint32_t)seed,
                               (uint32_t)(seed >> 32));
    float v[4];
    if (kind == 0) {
      box_muller(r.x, r.y, v[0], v[1]);
      box_muller(r.z, r.w, v[2], v[3]);
    } else if (kind == 1) {
      // [0, 1): the top 24 bits, so that 1.0 cannot come out (torch's uniform_ excludes the upper end too)
      v[0] = (float)(r.x >> 8) * 5.9604644775390625e-8f;
      v[1] = (float)(r.y >> 8) * 5.9604644775390625e-8f;
      v[2] = (float)(r.z >> 8) * 5.9604644775390625e-8f;
      v[3] = (float)(r.w >> 8) * 5.9604644775390625e-8f;
    } else {
      v[0] = __uint_as_float(r.x), v[1] = __uint_as_float(r.y), v[2] = __uint_as_float(r.z), v[3] = __uint_as_float(r.w);
    }
    const int64_t e = q << 2;
    if (e + 4 <= per && (per & 3) == 0) {
      st4(o + e, make_float4(v[0], v[1], v[2], v[3]));
    } else {
      for (int j = 0; j < 4 && e + j < per; ++j) o[e + j] = v[j];
    }
  }
  // every workgroup has read the draw index before it takes a ticket; the last one to arrive advances it and puts the
  // ticket counter back, both visible to the next launch on the stream
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
    const unsigned long long t = atomicAdd(&state[2], 1ull);
    if (t == total - 1) {
      state[2] = 0;
      state[1] = draw + 1;
    }
  }
}

// the class-dropout mask of CFG:84-90 in one launch: keep[b] = uniform(seed, sample id, draw, element 0) < prob, as uint8 —
// the value dmh_rng_indexed(kind 1) draws for the row, compared on the device (one block; B <= 65535)
__global__ __launch_bounds__(256) void rng_keep_mask_kernel(unsigned char* __restrict__ keep, int B, const int64_t* __restrict__ ids,
                                                            unsigned long long* state, float prob) {
  const unsigned long long seed = state[0], draw = state[1];
  for (int b = threadIdx.x; b < B; b += 256) {
    const unsigned long long sid = (unsigned long long)ids[b];
    const U4 r = philox4x32_10(U4{0u, (uint32_t)draw, (uint32_t)sid, (uint32_t)(sid >> 32)}, (uint32_t)seed, (uint32_t)(seed >> 32));
    keep[b] = ((float)(r.x >> 8) * 5.9604644775390625e-8f) < prob ? 1 : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) state[1] = draw + 1;
}

}  // namespace

extern "C" int dmh_rng_keep_mask(uint8_t* keep, int B, const int64_t* sample_ids, uint64_t* state, float prob, void* stream) {
  DMH_REQUIRE(keep && sample_ids && state && B > 0 && B <= 65535, "dmh_rng_keep_mask: bad arguments");
  hipLaunchKernelGGL(rng_keep_mask_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, keep, B, sample_ids,
                     (unsigned long long*)state, prob);
  DMH_CHECK_LAUNCH("dmh_rng_keep_mask");
  return DMH_OK;
}

extern "C" int dmh_rng_indexed(float* out, int B, int64_t per_sample, const int64_t* sample_ids, uint64_t* state, int kind,
                               void* stream) {
  DMH_REQUIRE(out && sample_ids && state && B > 0 && per_sample > 0, "dmh_rng_indexed: bad arguments");
  DMH_REQUIRE(kind >= 0 && kind <= 2, "dmh_rng_indexed: kind=%d (0 normal, 1 uniform, 2 raw words)", kind);
  DMH_REQUIRE(B <= 65535, "dmh_rng_indexed: B=%d (limit 65535 rows per launch)", B);
  DMH_REQUIRE(per_sample < ((int64_t)1 << 34), "dmh_rng_indexed: %lld elements per sample (limit 2^34)", (long long)per_sample);
  const int64_t nq = (per_sample + 3) >> 2;
  // every workgroup ends with a returning atomic on ONE word (the arrival ticket): at 2400 workgroups those tickets were most
  // of the launch (33 us for 2.4 M normals).  About two workgroups per CU in total, each looping over its share
  int64_t gx = cdiv64(nq, 256);
  const int64_t cap = B >= 512 ? 1 : 512 / B;
  gx = gx < 1 ? 1 : (gx > cap ? cap : gx);
  hipLaunchKernelGGL(rng_indexed_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, out, per_sample,
                     sample_ids, (unsigned long long*)state, kind);
  DMH_CHECK_LAUNCH("dmh_rng_indexed");
  return DMH_OK;
}
