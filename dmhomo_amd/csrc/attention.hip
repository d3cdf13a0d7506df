// K3 LinearAttention core (CFG:258-269) and K4 Attention core (CFG:287-295).
//
// qkv is the NHWC output of the to_qkv 1x1 conv: [B][n][384], channel = part*128 + head*32 + d
// (part 0 q, 1 k, 2 v; 4 heads x 32), so one pixel's q/k/v rows of a head are 128 contiguous
// bytes.  All contractions run on v_mfma_f32_32x32x2_f32 (exact fp32), one wave per head.
//
// Operand maps of the 32x32x2 MFMA (lane l: i = l&31, half = l>>5):
//   A[i][k=half], B[k=half][j=i];  D: col = l&31, row = (r&3) + 8*(r>>2) + 4*half, r = 0..15.
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define LA_NS 128                  // pixels per split of linattn pass 1
#define LA_PART (32 + 32 + 1024)   // per (b, split, head): max[32], sum[32], ctx[32][32]

// ------------------------------------------------------------------------------------------
// pass 1: per split, per head: m_d = max_n k[d,n]; p = exp(k - m); s_d = sum p; ctx[d][e] = sum_n p[d,n] v[e,n]
// (softmax over the n pixels is finished in the merge).  The split's k values live in registers.
__global__ __launch_bounds__(256) void linattn_context_kernel(const float* __restrict__ qkv,
                                                              float* __restrict__ partial, int n, int nsplit,
                                                              const int32_t* __restrict__ rows) {
  const int jb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  if (rows && jb >= rows[0]) return;   // (a row subset, common.h: the workgroups of inactive rows retire)
  const int b = dmh_rows_phys(rows, jb);
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int d = lane & 31, half = lane >> 5;
  const int p0 = sp * LA_NS;
  const float* base = qkv + (size_t)b * n * 384;

  // the split's k AND v values are all requested up front (128 independent loads in flight per wave) and
  // live in registers: the matrix loop below then never waits on memory
  float kr[LA_NS / 2], vr[LA_NS / 2];
  const float* kp = base + (size_t)p0 * 384 + 128 + h * 32 + d + (size_t)half * 384;
#pragma unroll
  for (int i = 0; i < LA_NS / 2; ++i) {
    const int pix = p0 + 2 * i + half;
    const bool ok = pix < n;
    const float* q = ok ? kp + (size_t)i * 768 : kp;  // clamped address, masked value: no branch around a load
    const float kv = q[0], vv = q[128];
    kr[i] = ok ? kv : -INFINITY;
    vr[i] = ok ? vv : 0.f;  // here d plays e
  }
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < LA_NS / 2; ++i) m = fmaxf(m, kr[i]);
  m = fmaxf(m, __shfl_xor(m, 32));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LA_NS / 2; ++i) {
    kr[i] = __expf(kr[i] - m);
    s += kr[i];
  }
  s += __shfl_xor(s, 32);

  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int i = 0; i < LA_NS / 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i], vr[i], acc, 0, 0, 0);
  float* out = partial + ((size_t)(b * nsplit + sp) * 4 + h) * LA_PART;
  if (half == 0) {
    out[d] = m;
    out[32 + d] = s;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;  // d
    out[64 + row * 32 + d] = acc[r];                     // col = e
  }
}

// merge the splits: M = max m; S = sum s*exp(m-M); ctx = sum ctx*exp(m-M) / S / n
__global__ __launch_bounds__(1024) void linattn_merge_kernel(const float* __restrict__ partial,
                                                             float* __restrict__ ctx, int n, int nsplit,
                                                             float* __restrict__ ms, const int32_t* __restrict__ rows) {
  if (rows && (int)(blockIdx.x >> 2) >= rows[0]) return;   // (a row subset, common.h)
  const int b = dmh_rows_phys(rows, blockIdx.x >> 2), h = blockIdx.x & 3;
  const int bh = b * 4 + h;
  const int d = threadIdx.x >> 5, e = threadIdx.x & 31;
  const float* base = partial + ((size_t)b * nsplit * 4 + h) * LA_PART;
  const size_t stride = (size_t)4 * LA_PART;
  float M = -INFINITY;
#pragma unroll 8  // the loads of 8 splits in flight (the chain itself is short; the latency of a dependent load per split was not)
  for (int sp = 0; sp < nsplit; ++sp) M = fmaxf(M, base[sp * stride + d]);
  float S = 0.f, acc = 0.f;
#pragma unroll 8
  for (int sp = 0; sp < nsplit; ++sp) {
    const float w = expf(base[sp * stride + d] - M);
    S = fmaf(base[sp * stride + 32 + d], w, S);
    acc = fmaf(base[sp * stride + 64 + d * 32 + e], w, acc);
  }
  ctx[(size_t)bh * 1024 + d * 32 + e] = acc / S / (float)n;
  if (ms && e == 0) {  // saved for the backward pass: softmax-over-n statistics of k
    ms[((size_t)bh * 32 + d) * 2 + 0] = M;
    ms[((size_t)bh * 32 + d) * 2 + 1] = S;
  }
}

// pass 2: out[p][h*32+e] = sum_d ctx[d][e] * q'[p][d],  q' = softmax_d(q[p]) * scale.
// K-slot map: lane half h, step s  <->  d = 16*half + s  (each lane loads 16 contiguous floats).
#define LA_TILES 4  // 32-pixel tiles per wave
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ ctx, float* __restrict__ out,
                                                            int n, int nblk, float scale,
                                                            const int32_t* __restrict__ rows) {
  const int jb = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  if (rows && jb >= rows[0]) return;   // (a row subset, common.h)
  const int b = dmh_rows_phys(rows, jb);
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int i = lane & 31, half = lane >> 5;
  const float* cb = ctx + ((size_t)(b * 4 + h)) * 1024;
  float bf[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) bf[s] = cb[(half * 16 + s) * 32 + i];  // B[k][j=e=i]
  const float* base = qkv + (size_t)b * n * 384;
  float* ob = out + (size_t)b * n * 128;
  for (int tI = 0; tI < LA_TILES; ++tI) {
    const int p0 = (blk * LA_TILES + tI) * 32;
    if (p0 >= n) break;
    const int pix = p0 + i;
    float q[16];
    if (pix < n) {
      const float* qp = base + (size_t)pix * 384 + h * 32 + half * 16;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 t = ld4(qp + s4 * 4);
        q[s4 * 4 + 0] = t.x;
        q[s4 * 4 + 1] = t.y;
        q[s4 * 4 + 2] = t.z;
        q[s4 * 4 + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) q[s] = 0.f;
    }
    float m = q[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) m = fmaxf(m, q[s]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      q[s] = expf(q[s] - m);
      sum += q[s];
    }
    sum += __shfl_xor(sum, 32);
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = (q[s] / sum) * scale;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bf[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;  // pixel in tile
      if (p0 + row < n) ob[(size_t)(p0 + row) * 128 + h * 32 + i] = acc[r];
    }
  }
}

// ------------------------------------------------------------------------------------------
// K4: softmax_j((q*scale)^T k) v, flash style with S^T = K Q^T so the softmax axis (keys) lies
// on accumulator rows (registers + lane half) and P^T is already in B-operand position for
// out^T[e][query] += V^T[e][key] P^T[key][query].  One wave per (sample, head, 32 queries).
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n,
                                                        int qtiles, float scale, const int32_t* __restrict__ rows) {
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, half = lane >> 5;
  const int qt = wid % qtiles;
  const int bh = wid / qtiles;
  const int h = bh & 3;
  if (rows && (bh >> 2) >= rows[0]) return;   // (a row subset, common.h; waves are independent: no workgroup barrier below)
  const int b = dmh_rows_phys(rows, bh >> 2);
  const float* base = qkv + (size_t)b * n * 384;
  const int q0 = qt * 32;

  // B operand of S^T: q[query=i][d = 16*half + s] * scale
  float qf[16];
  {
    const int qi = q0 + i;
    if (qi < n) {
      const float* qp = base + (size_t)qi * 384 + h * 32 + half * 16;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 t = ld4(qp + s4 * 4);
        qf[s4 * 4 + 0] = t.x * scale;
        qf[s4 * 4 + 1] = t.y * scale;
        qf[s4 * 4 + 2] = t.z * scale;
        qf[s4 * 4 + 3] = t.w * scale;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) qf[s] = 0.f;
    }
  }
  floatx16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  for (int k0 = 0; k0 < n; k0 += 32) {
    // A operand of S^T: k[key=i][d = 16*half + s]
    float kf[16];
    const int ki = k0 + i;
    if (ki < n) {
      const float* kp = base + (size_t)ki * 384 + 128 + h * 32 + half * 16;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 t = ld4(kp + s4 * 4);
        kf[s4 * 4 + 0] = t.x;
        kf[s4 * 4 + 1] = t.y;
        kf[s4 * 4 + 2] = t.z;
        kf[s4 * 4 + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) kf[s] = 0.f;
    }
    floatx16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sacc, 0, 0, 0);
    // sacc[r] = S^T[key = k0 + row(r)][query = i]
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= n) sacc[r] = -INFINITY;
      tmax = fmaxf(tmax, sacc[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = expf(mrun - mnew);  // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sacc[r] = expf(sacc[r] - mnew);
      psum += sacc[r];
    }
    psum += __shfl_xor(psum, 32);
    lrun = fmaf(lrun, alpha, psum);
    mrun = mnew;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    // out^T[e][query] += V^T[e][key] * P^T[key][query]; step r: key = k0 + row(r) for this lane half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float vv = (key < n) ? base[(size_t)key * 384 + 256 + h * 32 + i] : 0.f;  // A[i=e][k]
      oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, sacc[r], oacc, 0, 0, 0);
    }
  }
  // oacc[r] = out^T[e = row(r)][query = i]; rows 4g..4g+3 are contiguous e -> float4 stores
  const int qi = q0 + i;
  if (qi < n) {
    float* op = out + ((size_t)b * n + qi) * 128 + h * 32;
    const float inv = 1.0f / lrun;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      float4 o;
      o.x = oacc[g4 * 4 + 0] * inv;
      o.y = oacc[g4 * 4 + 1] * inv;
      o.z = oacc[g4 * 4 + 2] * inv;
      o.w = oacc[g4 * 4 + 3] * inv;
      st4(op + 8 * g4 + 4 * half, o);
    }
  }
}

// ------------------------------------------------------------------------------------------
extern "C" int dmh_linattn_splits(int n) { return dmh_dims_ok({n}, 1, 1 << 26) ? cdiv(n, LA_NS) : -1; }

extern "C" int64_t dmh_linattn_partial_floats(int B, int n) {
  if (!dmh_dims_ok({B}) || !dmh_dims_ok({n}, 1, 1 << 26)) return -1;
  return (int64_t)B * dmh_linattn_splits(n) * 4 * LA_PART;
}

extern "C" int dmh_linattn_context(const float* qkv, float* partial, int B, int n, const int32_t* rows, void* stream) {
  DMH_REQUIRE(qkv && partial && B > 0 && n > 0, "dmh_linattn_context: bad arguments");
  const int ns = dmh_linattn_splits(n);
  hipLaunchKernelGGL(linattn_context_kernel, dim3(B * ns), dim3(256), 0, (hipStream_t)stream, qkv, partial, n, ns, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_context");
  return DMH_OK;
}

extern "C" int dmh_linattn_merge(const float* partial, float* ctx, int B, int n, const int32_t* rows, void* stream) {
  DMH_REQUIRE(partial && ctx && B > 0 && n > 0, "dmh_linattn_merge: bad arguments");
  hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * 4), dim3(1024), 0, (hipStream_t)stream, partial, ctx, n,
                     dmh_linattn_splits(n), (float*)nullptr, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_merge");
  return DMH_OK;
}

// same merge for partials produced with another split count (linattn_fused.hip)
extern "C" int dmh_linattn_merge_n(const float* partial, float* ctx, int B, int n, int nsplit, const int32_t* rows,
                                   void* stream) {
  DMH_REQUIRE(partial && ctx && B > 0 && n > 0 && nsplit > 0, "dmh_linattn_merge_n: bad arguments");
  hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * 4), dim3(1024), 0, (hipStream_t)stream, partial, ctx, n, nsplit,
                     (float*)nullptr, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_merge_n");
  return DMH_OK;
}

// dmh_linattn_merge that also saves ms [B][4][32][2] = (max, sum exp) of k over the pixels, for dmh_linattn_backward
extern "C" int dmh_linattn_merge_ms(const float* partial, float* ctx, float* ms, int B, int n, void* stream) {
  DMH_REQUIRE(partial && ctx && ms && B > 0 && n > 0, "dmh_linattn_merge_ms: bad arguments");
  hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * 4), dim3(1024), 0, (hipStream_t)stream, partial, ctx, n,
                     dmh_linattn_splits(n), ms, (const int32_t*)nullptr);
  DMH_CHECK_LAUNCH("dmh_linattn_merge_ms");
  return DMH_OK;
}

extern "C" int dmh_linattn_apply(const float* qkv, const float* ctx, float* out, int B, int n, float scale,
                                 const int32_t* rows, void* stream) {
  DMH_REQUIRE(qkv && ctx && out && B > 0 && n > 0, "dmh_linattn_apply: bad arguments");
  const int nblk = cdiv(n, 32 * LA_TILES);
  hipLaunchKernelGGL(linattn_apply_kernel, dim3(B * nblk), dim3(256), 0, (hipStream_t)stream, qkv, ctx, out, n, nblk,
                     scale, rows);
  DMH_CHECK_LAUNCH("dmh_linattn_apply");
  return DMH_OK;
}

extern "C" int dmh_attention(const float* qkv, float* out, int B, int n, float scale, const int32_t* rows, void* stream) {
  DMH_REQUIRE(qkv && out && B > 0 && n > 0, "dmh_attention: bad arguments");
  const int qtiles = cdiv(n, 32);
  const int waves = B * 4 * qtiles;  // always a multiple of 4: one workgroup = the 4 heads' waves in flight
  hipLaunchKernelGGL(attention_kernel, dim3(waves / 4), dim3(256), 0, (hipStream_t)stream, qkv, out, n, qtiles, scale, rows);
  DMH_CHECK_LAUNCH("dmh_attention");
  return DMH_OK;
}
