// Training: a strided, batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for
// the SMALL products of the backward pass (the bottleneck Attention's n x n maps at n = 256..1024, the embedding MLPs),
// and row softmax forward / backward.  One wave per 32x32 output tile, operands read straight from global memory
// (their lines stay in L1 across the K loop) — adequate for these sizes, not a general GEMM.
//   C[bo][bi][m][n] = alpha * sum_k A[bo][bi][m][k] * B[bo][bi][k][n]      every operand with its own 4 strides
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmOperand {
  const float* p;
  int64_t s_bo, s_bi, s_r, s_c;  // strides (floats) of: outer batch, inner batch, row, column
};

__global__ __launch_bounds__(64) void bgemm_kernel(GemmOperand A, GemmOperand Bm, float* __restrict__ C, int64_t c_bo,
                                                   int64_t c_bi, int64_t c_r, int64_t c_c, int M, int N, int K, int nbi,
                                                   float alpha) {
  const int tilesN = (N + 31) / 32;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
  const int bo = blockIdx.y / nbi, bi = blockIdx.y % nbi;
  const int lane = threadIdx.x, i = lane & 31, half = lane >> 5;
  const float* a = A.p + bo * A.s_bo + bi * A.s_bi;
  const float* b = Bm.p + bo * Bm.s_bo + bi * Bm.s_bi;
  const int m = tm * 32 + i, n = tn * 32 + i;
  const bool mok = m < M, nok = n < N;
  const float* ar = a + (int64_t)(mok ? m : 0) * A.s_r;
  const float* bc = b + (int64_t)(nok ? n : 0) * Bm.s_c;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 2) {
    const int k = k0 + half;
    const bool kok = k < K;
    const float av = (mok && kok) ? ar[(int64_t)k * A.s_c] : 0.f;
    const float bv = (nok && kok) ? bc[(int64_t)k * Bm.s_r] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
  float* c = C + bo * c_bo + bi * c_bi;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row < M && nok) c[(int64_t)row * c_r + (int64_t)n * c_c] = acc[r] * alpha;
  }
}

// P[row][j] = softmax_j(S[row][j]); one wave per row of length n (contiguous)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, float* __restrict__ P, int64_t rows,
                                                           int n) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* s = S + row * n;
  float m = -INFINITY;
  for (int j = lane; j < n; j += 64) m = fmaxf(m, s[j]);
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) sum += expf(s[j] - m);
  for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off);
  for (int j = lane; j < n; j += 64) P[row * n + j] = expf(s[j] - m) / sum;
}

// dS = P * (dP - sum_j dP*P), in place on dP
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                               int64_t rows, int n) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + row * n;
  float* d = dP + row * n;
  float dot = 0.f;
  for (int j = lane; j < n; j += 64) dot = fmaf(d[j], p[j], dot);
  for (int off = 32; off; off >>= 1) dot += __shfl_xor(dot, off);
  for (int j = lane; j < n; j += 64) d[j] = p[j] * (d[j] - dot);
}

// strides arrays: {outer batch, inner batch, row, column} in floats
extern "C" int dmh_bgemm(const float* A, const int64_t* sa, const float* B, const int64_t* sb, float* C, const int64_t* sc,
                         int M, int N, int K, int nbo, int nbi, float alpha, void* stream) {
  DMH_REQUIRE(A && B && C && sa && sb && sc, "dmh_bgemm: null pointer");
  DMH_REQUIRE(M > 0 && N > 0 && K > 0 && nbo > 0 && nbi > 0 && (int64_t)nbo * nbi < 65536, "dmh_bgemm: bad shape");
  GemmOperand a = {A, sa[0], sa[1], sa[2], sa[3]}, b = {B, sb[0], sb[1], sb[2], sb[3]};
  dim3 grid(cdiv(M, 32) * cdiv(N, 32), nbo * nbi);
  hipLaunchKernelGGL(bgemm_kernel, grid, dim3(64), 0, (hipStream_t)stream, a, b, C, sc[0], sc[1], sc[2], sc[3], M, N, K, nbi,
                     alpha);
  DMH_CHECK_LAUNCH("dmh_bgemm");
  return DMH_OK;
}

extern "C" int dmh_softmax_rows(const float* S, float* P, int64_t rows, int n, void* stream) {
  DMH_REQUIRE(S && P && rows > 0 && n > 0, "dmh_softmax_rows: bad arguments");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, S, P, rows, n);
  DMH_CHECK_LAUNCH("dmh_softmax_rows");
  return DMH_OK;
}

extern "C" int dmh_softmax_rows_backward(const float* P, float* dP, int64_t rows, int n, void* stream) {
  DMH_REQUIRE(P && dP && rows > 0 && n > 0, "dmh_softmax_rows_backward: bad arguments");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, P, dP,
                     rows, n);
  DMH_CHECK_LAUNCH("dmh_softmax_rows_backward");
  return DMH_OK;
}
