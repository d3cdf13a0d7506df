// Training: backward of the LinearAttention core (CFG:258-269) on a
// stored qkv tensor [B][n][384] (channel = part*128 + head*32 + d), as the unfused forward of attention.hip computes it:
//   q' = softmax_d(q) * scale     k' = softmax_n(k)     ctx[d,e] = sum_n k'[d,n] v[e,n] / n     out[e,n] = sum_d ctx[d,e] q'[d,n]
// Given dO = d out:
//   dctx[d,e] = sum_n q'[d,n] dO[e,n]                     (reduction over pixels: pixel_outer_kernel + split sum)
//   dq'[d,n]  = sum_e ctx[d,e] dO[e,n]      dq = s*(ds - sum_d ds*s),  s = softmax_d(q), ds = dq'*scale
//   dk'[d,n]  = sum_e dctx[d,e] v[e,n] / n  dv[e,n] = sum_d dctx[d,e] k'[d,n] / n
//   dk[d,n]   = k'[d,n] * (dk'[d,n] - t[d]),   t[d] = sum_n k'[d,n] dk'[d,n]          (softmax over the n pixels)
// All 32x32 products run on v_mfma_f32_32x32x2_f32 in the "channels x pixels" orientation: the accumulator holds, per
// lane, one pixel column and 16 of its 32 channels (the other 16 sit in lane ^ 32), so every per-pixel reduction over
// channels is 16 in-lane adds and one shuffle.  One wave per head.
//
// Operand maps of the 32x32x2 MFMA (lane l: i = l&31, half = l>>5):
//   A[i][k=half], B[k=half][j=i];  D: col = l&31, row = (r&3) + 8*(r>>2) + 4*half, r = 0..15.
// K-slot map over the 16 steps s: lane half h, step s  <->  channel 16*h + s  (16 contiguous floats per lane).
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define LAB_TILES 4  // 32-pixel tiles per wave

namespace {
// acc[row c][col pixel] = sum_k mat[c][k] * x[pixel][k]:  a16[s] = mat[c = i][k = 16*half + s], x16[s] = x[pixel i][16*half + s]
__device__ __forceinline__ floatx16 mat_times_xT(const float (&a16)[16], const float (&x16)[16]) {
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a16[s], x16[s], acc, 0, 0, 0);
  return acc;
}
// 16 contiguous floats of a pixel row (or zeros)
__device__ __forceinline__ void load16(float (&v)[16], const float* p, bool ok) {
  if (ok) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float4 t = ld4(p + s4 * 4);
      v[s4 * 4 + 0] = t.x;
      v[s4 * 4 + 1] = t.y;
      v[s4 * 4 + 2] = t.z;
      v[s4 * 4 + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = 0.f;
  }
}
// the 16 accumulator rows of this lane are channels 4*half + 8*g + (0..3), g = 0..3: four float4 at base + 4*half + 8*g
__device__ __forceinline__ void loadD(floatx16& v, const float* base, int half, bool ok) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) t = ld4(base + 4 * half + 8 * g);
    v[g * 4 + 0] = t.x;
    v[g * 4 + 1] = t.y;
    v[g * 4 + 2] = t.z;
    v[g * 4 + 3] = t.w;
  }
}
__device__ __forceinline__ void storeD(const floatx16& v, float* base, int half, bool ok) {
  if (!ok) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) st4(base + 4 * half + 8 * g, make_float4(v[g * 4 + 0], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]));
}
}  // namespace

// ---- A1: qs = softmax_d(q)*scale (kept for the dctx reduction), dq written into dqkv[..][0..127]
__global__ __launch_bounds__(256) void linattn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                            const float* __restrict__ dout, float* __restrict__ qs,
                                                            float* __restrict__ dqkv, int n, int nblk, float scale) {
  const int b = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  const float* cb = ctx + ((size_t)(b * 4 + h)) * 1024;
  float ca[16];  // A[i = d][k = e = 16*half + s] = ctx[d][e]
#pragma unroll
  for (int s = 0; s < 16; ++s) ca[s] = cb[i * 32 + half * 16 + s];
  for (int tI = 0; tI < LAB_TILES; ++tI) {
    const int p0 = (blk * LAB_TILES + tI) * 32;
    if (p0 >= n) break;
    const int pix = p0 + i;
    const bool ok = pix < n;
    const size_t row = (size_t)b * n + (ok ? pix : 0);
    floatx16 q;
    loadD(q, qkv + row * 384 + h * 32, half, ok);
    float m = q[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, q[r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      q[r] = expf(q[r] - m);
      sum += q[r];
    }
    sum += __shfl_xor(sum, 32);
    floatx16 s16, qsv;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s16[r] = q[r] / sum;
      qsv[r] = s16[r] * scale;
    }
    storeD(qsv, qs + row * 128 + h * 32, half, ok);
    float d16[16];
    load16(d16, dout + row * 128 + h * 32 + half * 16, ok);
    floatx16 dqp = mat_times_xT(ca, d16);  // dq'[d][pixel]
    float dot = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dqp[r] *= scale;  // ds
      dot = fmaf(dqp[r], s16[r], dot);
    }
    dot += __shfl_xor(dot, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) dqp[r] = s16[r] * (dqp[r] - dot);
    storeD(dqp, dqkv + row * 384 + h * 32, half, ok);
  }
}

// ---- A2: part[split][b][h][d][e] = sum over the split's pixels of x[n][d] * y[n][e]   (x, y: [B][n][128], channel h*32+.)
#define LAB_NS 128
__global__ __launch_bounds__(256) void pixel_outer_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ part, int n, int nsplit, int B) {
  const int b = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int d = lane & 31, half = lane >> 5;
  const int p0 = sp * LAB_NS;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
  for (int i = 0; i < LAB_NS / 2; ++i) {
    const int pix = p0 + 2 * i + half;
    const bool ok = pix < n;
    const size_t row = (size_t)b * n + (ok ? pix : 0);
    const float xv = x[row * 128 + h * 32 + d], yv = y[row * 128 + h * 32 + d];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? xv : 0.f, ok ? yv : 0.f, acc, 0, 0, 0);
  }
  float* out = part + ((size_t)(sp * B + b) * 4 + h) * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + d] = acc[r];  // [row = x channel][col = y channel]
}

// ---- B: dqkv k part <- k' * dk'  (the softmax-over-n correction follows), dqkv v part <- dv
// ms: [B][4][32][2] = (max, sum exp) of k over the n pixels, from the forward merge
__global__ __launch_bounds__(256) void linattn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                             const float* __restrict__ ms, float* __restrict__ dqkv, int n,
                                                             int nblk) {
  const int b = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  const float* dc = dctx + ((size_t)(b * 4 + h)) * 1024;
  const float* msb = ms + ((size_t)(b * 4 + h)) * 64;
  const float inv_n = 1.0f / (float)n;
  float a_dk[16], a_dv[16];  // A operands: dctx[d = i][e = 16*half + s]  and  dctx^T: [e = i][d = 16*half + s]
  float mB[16], sB[16];      // k-softmax statistics in B-operand channel order d = 16*half + s
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    a_dk[s] = dc[i * 32 + half * 16 + s] * inv_n;
    a_dv[s] = dc[(half * 16 + s) * 32 + i] * inv_n;
    mB[s] = msb[(half * 16 + s) * 2 + 0];
    sB[s] = 1.0f / msb[(half * 16 + s) * 2 + 1];
  }
  floatx16 mD, sD;  // ... and in accumulator-row order d = 4*half + 8*g + j
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dd = (r & 3) + 8 * (r >> 2) + 4 * half;
    mD[r] = msb[dd * 2 + 0];
    sD[r] = 1.0f / msb[dd * 2 + 1];
  }
  for (int tI = 0; tI < LAB_TILES; ++tI) {
    const int p0 = (blk * LAB_TILES + tI) * 32;
    if (p0 >= n) break;
    const int pix = p0 + i;
    const bool ok = pix < n;
    const size_t row = (size_t)b * n + (ok ? pix : 0);
    float k16[16], v16[16];
    load16(k16, qkv + row * 384 + 128 + h * 32 + half * 16, ok);
    load16(v16, qkv + row * 384 + 256 + h * 32 + half * 16, ok);
#pragma unroll
    for (int s = 0; s < 16; ++s) k16[s] = ok ? expf(k16[s] - mB[s]) * sB[s] : 0.f;  // k'
    floatx16 dkp = mat_times_xT(a_dk, v16);  // dk'[d][pixel]
    floatx16 dv = mat_times_xT(a_dv, k16);   // dv[e][pixel]
    floatx16 kD;
    loadD(kD, qkv + row * 384 + 128 + h * 32, half, ok);
#pragma unroll
    for (int r = 0; r < 16; ++r) dkp[r] *= expf(kD[r] - mD[r]) * sD[r];  // k' * dk'
    storeD(dkp, dqkv + row * 384 + 128 + h * 32, half, ok);
    storeD(dv, dqkv + row * 384 + 256 + h * 32, half, ok);
  }
}

// ---- column sums over the pixels of channels [c0, c0 + 128) of a [B][n][ld] tensor: part[chunk][b][128]
__global__ __launch_bounds__(256) void colsum128_kernel(const float* __restrict__ x, float* __restrict__ part, int n, int ld,
                                                        int c0, int nchunk, int B) {
  __shared__ float red[256 * 4];
  const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
  const int q = threadIdx.x & 31, pl = threadIdx.x >> 5;  // 32 channel quads x 8 pixel lanes
  const int p0 = ch * 256, p1 = min(p0 + 256, n);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = p0 + pl; p < p1; p += 8) {
    const float4 v = ld4(x + ((size_t)b * n + p) * ld + c0 + q * 4);
    s.x += v.x;
    s.y += v.y;
    s.z += v.z;
    s.w += v.w;
  }
  st4(red + threadIdx.x * 4, s);
  __syncthreads();
  if (pl == 0) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < 8; ++l) {
      const float4 v = ld4(red + (l * 32 + q) * 4);
      a.x += v.x;
      a.y += v.y;
      a.z += v.z;
      a.w += v.w;
    }
    st4(part + ((size_t)ch * B + b) * 128 + q * 4, a);
  }
}

// ---- C: dk = (k' dk') - k' * t   in place on the k part of dqkv;  t: [B][128] (channel h*32 + d)
__global__ __launch_bounds__(256) void linattn_bwd_kfix_kernel(const float* __restrict__ qkv, const float* __restrict__ ms,
                                                               const float* __restrict__ t, float* __restrict__ dqkv,
                                                               int64_t npix, int n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one float4 of the 128 k channels of a pixel
  if (i >= npix * 32) return;
  const int64_t pixg = i >> 5;
  const int c = (int)(i & 31) * 4;
  const int b = (int)(pixg / n);
  const float4 kv = ld4(qkv + pixg * 384 + 128 + c);
  float4 g = ld4(dqkv + pixg * 384 + 128 + c);
  const float* msb = ms + ((size_t)b * 128 + c) * 2;
  const float4 tv = ld4(t + (size_t)b * 128 + c);
  g.x -= expf(kv.x - msb[0]) / msb[1] * tv.x;
  g.y -= expf(kv.y - msb[2]) / msb[3] * tv.y;
  g.z -= expf(kv.z - msb[4]) / msb[5] * tv.z;
  g.w -= expf(kv.w - msb[6]) / msb[7] * tv.w;
  st4(dqkv + pixg * 384 + 128 + c, g);
}

// sum of `count` consecutive slabs of `per` floats (fixed order)
__global__ void sum_slabs_kernel(const float* __restrict__ in, float* __restrict__ out, int count, int64_t per) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  float s = 0.f;
  for (int k = 0; k < count; ++k) s += in[(size_t)k * per + i];
  out[i] = s;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int64_t dmh_linattn_bwd_workspace_floats(int B, int n) {
  if (!dmh_dims_ok({B}) || !dmh_dims_ok({n}, 1, 1 << 26)) return -1;
  const int ns = cdiv(n, LAB_NS), nc = cdiv(n, 256);
  return (int64_t)B * n * 128                 // qs
         + (int64_t)ns * B * 4 * 1024         // dctx partials
         + (int64_t)B * 4 * 1024              // dctx
         + (int64_t)nc * B * 128              // t partials
         + (int64_t)B * 128;                  // t
}

// qkv [B][n][384], ctx [B][4][32][32] and ms [B][4][32][2] from the forward (dmh_linattn_merge_ms), dout [B][n][128]
// -> dqkv [B][n][384].  work: dmh_linattn_bwd_workspace_floats floats.
extern "C" int dmh_linattn_backward(const float* qkv, const float* ctx, const float* ms, const float* dout, float* dqkv,
                                    float* work, int B, int n, float scale, void* stream) {
  DMH_REQUIRE(qkv && ctx && ms && dout && dqkv && work && B > 0 && n > 0, "dmh_linattn_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int ns = cdiv(n, LAB_NS), nc = cdiv(n, 256), nblk = cdiv(n, 32 * LAB_TILES);
  float* qs = work;
  float* dpart = qs + (int64_t)B * n * 128;
  float* dctx = dpart + (int64_t)ns * B * 4 * 1024;
  float* tpart = dctx + (int64_t)B * 4 * 1024;
  float* t = tpart + (int64_t)nc * B * 128;
  hipLaunchKernelGGL(linattn_bwd_q_kernel, dim3(B * nblk), dim3(256), 0, st, qkv, ctx, dout, qs, dqkv, n, nblk, scale);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(q)");
  hipLaunchKernelGGL(pixel_outer_kernel, dim3(B * ns), dim3(256), 0, st, qs, dout, dpart, n, ns, B);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(dctx)");
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv64((int64_t)B * 4096, 256)), dim3(256), 0, st, dpart, dctx, ns,
                     (int64_t)B * 4096);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(dctx sum)");
  hipLaunchKernelGGL(linattn_bwd_kv_kernel, dim3(B * nblk), dim3(256), 0, st, qkv, dctx, ms, dqkv, n, nblk);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(kv)");
  hipLaunchKernelGGL(colsum128_kernel, dim3(B * nc), dim3(256), 0, st, dqkv, tpart, n, 384, 128, nc, B);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(t)");
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv64((int64_t)B * 128, 256)), dim3(256), 0, st, tpart, t, nc,
                     (int64_t)B * 128);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(t sum)");
  hipLaunchKernelGGL(linattn_bwd_kfix_kernel, dim3((unsigned)cdiv64((int64_t)B * n * 32, 256)), dim3(256), 0, st, qkv, ms,
                     t, dqkv, (int64_t)B * n, n);
  DMH_CHECK_LAUNCH("dmh_linattn_backward(k fix)");
  return DMH_OK;
}
