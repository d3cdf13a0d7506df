// K5 — time / class embeddings and the small dense layers (latency-bound, a few µs each).
#include "common.h"

// N7: out[r][i] = sin(t_r * f_i), out[r][half+i] = cos(t_r * f_i)   (t int64 -> fp32, CFG:170-171)
__global__ void sinusoidal_embed_kernel(const int64_t* __restrict__ t, const float* __restrict__ freq,
                                        float* __restrict__ out, int R, int dim) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (idx >= R * half) return;
  const int r = idx / half, i = idx % half;
  const float arg = (float)t[r] * freq[i];
  out[(size_t)r * dim + i] = sinf(arg);
  out[(size_t)r * dim + half + i] = cosf(arg);
}

// RandomOrLearnedSinusoidalPosEmb (CFG:185-190): out[r] = (t_r, sin(t_r w_i 2 pi) ..., cos(t_r w_i 2 pi) ...), the products in the
// reference's order ((t * w) * 2) * pi in fp32
__global__ void fourier_embed_kernel(const int64_t* __restrict__ t, const float* __restrict__ w, float* __restrict__ out, int R,
                                     int half) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * half) return;
  const int r = idx / half, i = idx % half;
  const float x = (float)t[r];
  const float arg = ((x * w[i]) * 2.0f) * 3.14159265358979323846f;
  float* o = out + (size_t)r * (2 * half + 1);
  if (i == 0) o[0] = x;
  o[1 + i] = sinf(arg);
  o[1 + half + i] = cosf(arg);
}

// N8: classes_emb(classes) with rows swapped for null_classes_emb where keep == 0 (CFG:419-425)
__global__ void class_embed_kernel(const int64_t* __restrict__ classes, const uint8_t* __restrict__ keep,
                                   const float* __restrict__ table, const float* __restrict__ null_emb,
                                   float* __restrict__ out, int R, int dim, int num_classes) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * dim) return;
  const int r = idx / dim, i = idx % dim;
  const bool k = keep ? (keep[r] != 0) : true;
  const int64_t c = classes[r];
  // (a class id outside the table is device data the launch cannot refuse — nn.Embedding raises on the host / asserts on the
  //  device there, CFG:419: the row becomes NaN instead of whatever lies beside the table)
  out[idx] = k ? ((c >= 0 && c < num_classes) ? table[(size_t)c * dim + i] : __builtin_nanf("")) : null_emb[i];
}

__device__ __forceinline__ float act_f(float x, int act) {
  if (act == 1) return silu_f(x);
  if (act == 2) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));  // exact (erf) GELU
  return x;
}

// y[r][o] = act_out(sum_i act_in(x[r][i]) * wt[i][o] + bias[o]); RB rows per blockIdx.y, 64 output features per
// blockIdx.x, the (activated) input rows staged in LDS.  The four waves of a block each take a quarter of the input features
// (round 2: one thread per output walked all of them — a dependent chain of in_dim loads + FMAs, 14 us per launch for a few
// MFLOP, a dozen launches at the head of every UNet pass); the quarters are added in a fixed order.  A weight is loaded once
// for the block's RB rows (the (scale, shift) projection of all ResnetBlocks, 512 -> 9.5 k features, re-read its 19 MB of
// weights from L2 for every row: 57 us at 50 rows); every row keeps its own accumulator and summation order, so a row's
// result does not depend on which rows share its block.
constexpr int LIN_RB = 4;
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, int64_t x_stride,
                                                     const float* __restrict__ wt, const float* __restrict__ bias,
                                                     float* __restrict__ y, int64_t y_stride, int R, int in_dim, int out_dim,
                                                     int act_in, int act_out) {
  extern __shared__ float xs[];          // LIN_RB x in_dim staged inputs, then LIN_RB x 4 x 64 partial sums
  float* red = xs + LIN_RB * in_dim;
  const int r0 = blockIdx.y * LIN_RB;
#pragma unroll
  for (int rr = 0; rr < LIN_RB; ++rr) {
    const int r = min(r0 + rr, R - 1);   // rows past the end repeat the last one and are not stored
    for (int i = threadIdx.x; i < in_dim; i += 256) xs[rr * in_dim + i] = act_f(x[r * x_stride + i], act_in);
  }
  __syncthreads();
  const int ol = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + ol;
  const int oc = o < out_dim ? o : out_dim - 1;      // clamped column: every lane walks its quarter, the store is guarded
  const int kq = (in_dim + 3) >> 2;
  const int k0 = ks * kq, k1 = min(k0 + kq, in_dim);
  float acc[LIN_RB];
#pragma unroll
  for (int rr = 0; rr < LIN_RB; ++rr) acc[rr] = 0.f;
#pragma unroll 8
  for (int i = k0; i < k1; ++i) {
    const float w = wt[(size_t)i * out_dim + oc];
#pragma unroll
    for (int rr = 0; rr < LIN_RB; ++rr) acc[rr] = fmaf(xs[rr * in_dim + i], w, acc[rr]);
  }
#pragma unroll
  for (int rr = 0; rr < LIN_RB; ++rr) red[(rr * 4 + ks) * 64 + ol] = acc[rr];
  __syncthreads();
  if (o < out_dim && r0 + ks < R) {      // wave ks finishes row r0 + ks
    const float* q = red + (ks * 4) * 64 + ol;
    float v = (q[0] + q[64]) + (q[128] + q[192]);
    if (bias) v += bias[o];
    y[(r0 + ks) * y_stride + o] = act_f(v, act_out);
  }
}

// (scale, shift) rows of a replayed denoise step from tables: the linear over the concatenated ResnetBlock mlps is
// v = (q0 + q1) + (q2 + q3) + bias with q0, q1 the partial sums over the TIME half of its input and q2, q3 those over the CLASS
// half (linear_kernel, in_dim = 512 in four quarters) — the first pair depends on the timestep only, the second on (class, kept
// or dropped) only.  T[s] = q0 + q1 of denoise step s and C[c] = q2 + q3 of class c (row ncls: the null embedding) are made once
// per weight version and schedule by linear_kernel itself (the other half of its input zeroed: SiLU(0) = 0 adds exact zeros),
// and a step's rows are  out[b] = (T[*cursor] + C[keep[b] ? classes[b] : ncls]) + bias  — bitwise what the six launches at the
// head of a pass (two embeddings, four small linears, the 512 -> 8 k linear: ~87 us alone on the chip) compute.
__global__ __launch_bounds__(256) void ss_gather_kernel(const float* __restrict__ T, const float* __restrict__ Ct,
                                                        const float* __restrict__ bias, const int32_t* __restrict__ cursor,
                                                        const int64_t* __restrict__ classes, const unsigned char* __restrict__ keep,
                                                        int ncls, float* __restrict__ out, int N, int S) {
  const int b = blockIdx.y;
  const int o4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (o4 >= N) return;
  const int step = *cursor;
  const bool kept = keep ? keep[b] != 0 : true;
  const int64_t cls = classes[b];
  // both indices are device data the launch cannot check (a caller's class id, the replayed loop's cursor): one outside its
  // table poisons the row with NaN — a visibly broken sample — instead of reading whatever lies beside the table
  if ((unsigned)step >= (unsigned)S || (kept && (cls < 0 || cls >= ncls))) {
    const float q = __builtin_nanf("");
    st4(out + (size_t)b * N + o4, make_float4(q, q, q, q));
    return;
  }
  const int row = kept ? (int)cls : ncls;
  const float4 t = ld4(T + (size_t)step * N + o4), c = ld4(Ct + (size_t)row * N + o4), bb = ld4(bias + o4);
  st4(out + (size_t)b * N + o4, make_float4((t.x + c.x) + bb.x, (t.y + c.y) + bb.y, (t.z + c.z) + bb.z, (t.w + c.w) + bb.w));
}

extern "C" int dmh_ss_gather(const float* T, const float* C, const float* bias, const int32_t* cursor, const int64_t* classes,
                             const uint8_t* keep, int ncls, float* out, int B, int N, int S, void* stream) {
  DMH_REQUIRE(T && C && bias && cursor && classes && out && B > 0 && N > 0 && N % 4 == 0 && ncls > 0 && B <= 65535 && S > 0,
              "dmh_ss_gather: bad arguments");
  hipLaunchKernelGGL(ss_gather_kernel, dim3(cdiv(N / 4, 256), B), dim3(256), 0, (hipStream_t)stream, T, C, bias, cursor, classes,
                     keep, ncls, out, N, S);
  DMH_CHECK_LAUNCH("dmh_ss_gather");
  return DMH_OK;
}

extern "C" int dmh_sinusoidal_embed(const int64_t* t, const float* freq, float* out, int R, int dim, void* stream) {
  DMH_REQUIRE(t && freq && out && R > 0 && dim > 0 && dim % 2 == 0, "dmh_sinusoidal_embed: bad arguments");
  const int total = R * (dim / 2);
  hipLaunchKernelGGL(sinusoidal_embed_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, t, freq, out,
                     R, dim);
  DMH_CHECK_LAUNCH("dmh_sinusoidal_embed");
  return DMH_OK;
}

extern "C" int dmh_fourier_embed(const int64_t* t, const float* weights, float* out, int R, int half, void* stream) {
  DMH_REQUIRE(t && weights && out && R > 0 && half > 0, "dmh_fourier_embed: bad arguments");
  hipLaunchKernelGGL(fourier_embed_kernel, dim3(cdiv(R * half, 256)), dim3(256), 0, (hipStream_t)stream, t, weights, out, R,
                     half);
  DMH_CHECK_LAUNCH("dmh_fourier_embed");
  return DMH_OK;
}

extern "C" int dmh_class_embed(const int64_t* classes, const uint8_t* keep, const float* table, const float* null_emb,
                               float* out, int R, int dim, int num_classes, void* stream) {
  DMH_REQUIRE(classes && table && null_emb && out && R > 0 && dim > 0 && num_classes > 0, "dmh_class_embed: bad arguments");
  hipLaunchKernelGGL(class_embed_kernel, dim3(cdiv(R * dim, 256)), dim3(256), 0, (hipStream_t)stream, classes, keep,
                     table, null_emb, out, R, dim, num_classes);
  DMH_CHECK_LAUNCH("dmh_class_embed");
  return DMH_OK;
}

extern "C" int dmh_linear(const float* x, int64_t x_stride, const float* wt, const float* bias, float* y,
                          int64_t y_stride, int R, int in_dim, int out_dim, int act_in, int act_out, void* stream) {
  DMH_REQUIRE(x && wt && y && R > 0 && in_dim > 0 && out_dim > 0, "dmh_linear: bad arguments");
  DMH_REQUIRE(in_dim <= 3584, "dmh_linear: in_dim %d too large (the block stages 4 input rows in 60 KB of LDS)", in_dim);
  hipLaunchKernelGGL(linear_kernel, dim3(cdiv(out_dim, 64), cdiv(R, LIN_RB)), dim3(256),
                     LIN_RB * (in_dim + 256) * sizeof(float), (hipStream_t)stream, x, x_stride, wt, bias, y, y_stride, R,
                     in_dim, out_dim, act_in, act_out);
  DMH_CHECK_LAUNCH("dmh_linear");
  return DMH_OK;
}
