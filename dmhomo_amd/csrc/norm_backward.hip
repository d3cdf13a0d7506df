// Training: backward of Block = GroupNorm -> (scale+1, shift) -> SiLU
// (CFG:196-213) given the conv output y it normalised, and of the weight standardisation (CFG:120-126).
//
// Forward (as dmh_gn_finalize folds it): z = a*y + c with a = rstd*gamma*(s+1), c = (beta - mean*rstd*gamma)*(s+1) + t,
// out = SiLU(z).  With xh = (y - mean)*rstd, k = gamma*(s+1), dz = dout * SiLU'(z):
//   S1[b,c] = sum_hw dz          S2[b,c] = sum_hw dz*y          T2 = rstd*(S2 - mean*S1)  (= sum dz*xh)
//   m1[b,g] = mean_{c in g, hw} k*dz = sum_c k*S1 / N           m2[b,g] = sum_c k*T2 / N
//   dy = rstd*(k*dz - m1 - xh*m2)  =  P[b,c]*dz + Q[b,g]*y + R[b,g]
//   dgamma[c] = sum_b (s+1)*T2    dbeta[c] = sum_b (s+1)*S1    ds[b,c] = gamma*T2 + beta*S1    dt[b,c] = S1
// Three kernels: per-chunk partial sums (fixed order, no atomics) -> per (sample, group) finalize in f64 -> elementwise
// apply.  NHWC fp32.
#include "common.h"

#define GNB_PCH 256  // pixels per partial chunk

// SiLU'(z) = sg * (1 + z * (1 - sg)), sg = sigmoid(z) on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1-2 ulp) like
// the forward's silu_fast (round 3: the exact expf + IEEE division made the reduce / apply kernels vector-bound — two reads of
// the block's tensors at 3.7 TB/s)
__device__ __forceinline__ float silu_grad(float z) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

// part[b][chunk][c][2] = (sum dz, sum dz*y) over the chunk's pixels.  256 threads = (C/4 channel quads) x pixel lanes.
__global__ __launch_bounds__(256) void gn_silu_bwd_reduce_kernel(const float* __restrict__ dout,
                                                                 const float* __restrict__ y,
                                                                 const float* __restrict__ coef, float* __restrict__ part,
                                                                 int HW, int C, int nchunk) {
  __shared__ float red[256 * 8];
  const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
  const int C4 = C >> 2;
  const int tid = threadIdx.x;
  const int p0 = ch * GNB_PCH, p1 = min(p0 + GNB_PCH, HW);
  // quads are walked in groups of up to 256 / plan: each thread owns quad q (strided over C4) and pixel lane pl
  const int qpt = C4 < 256 ? C4 : 256;        // quads handled concurrently
  const int plan = 256 / qpt;                 // pixel lanes
  const int q0 = tid % qpt, pl = tid / qpt;
  for (int qb = 0; qb < C4; qb += qpt) {
    const int q = qb + q0;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (q < C4 && pl < plan) {
      const float4 a = ld4(coef + (size_t)(b * 2 + 0) * C + q * 4);
      const float4 cc = ld4(coef + (size_t)(b * 2 + 1) * C + q * 4);
      for (int p = p0 + pl; p < p1; p += plan) {
        const size_t off = ((size_t)b * HW + p) * C + q * 4;
        const float4 d = ld4(dout + off), v = ld4(y + off);
        const float dzx = d.x * silu_grad(fmaf(a.x, v.x, cc.x)), dzy = d.y * silu_grad(fmaf(a.y, v.y, cc.y));
        const float dzz = d.z * silu_grad(fmaf(a.z, v.z, cc.z)), dzw = d.w * silu_grad(fmaf(a.w, v.w, cc.w));
        s1.x += dzx;
        s1.y += dzy;
        s1.z += dzz;
        s1.w += dzw;
        s2.x = fmaf(dzx, v.x, s2.x);
        s2.y = fmaf(dzy, v.y, s2.y);
        s2.z = fmaf(dzz, v.z, s2.z);
        s2.w = fmaf(dzw, v.w, s2.w);
      }
    }
    float* r = red + tid * 8;
    r[0] = s1.x; r[1] = s1.y; r[2] = s1.z; r[3] = s1.w;
    r[4] = s2.x; r[5] = s2.y; r[6] = s2.z; r[7] = s2.w;
    __syncthreads();
    if (pl == 0 && q < C4) {  // fixed order over the pixel lanes
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int l = 0; l < plan; ++l)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += red[(l * qpt + q0) * 8 + j];
      float* o = part + (((size_t)b * nchunk + ch) * C + q * 4) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j * 2 + 0] = acc[j];
        o[j * 2 + 1] = acc[4 + j];
      }
    }
    __syncthreads();
  }
}

// one wave per (sample, group): bcoef[b][0][c] = P, [1][c] = Q, [2][c] = R; pg[b][0..3][c] = per-sample parts of
// dgamma, dbeta, and d(scale), d(shift)
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(const float* __restrict__ part, int nchunk,
                                                             const float* __restrict__ mr, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ ss,
                                                             int64_t ss_stride, float* __restrict__ bcoef,
                                                             float* __restrict__ pg, int C, int groups, int hw) {
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cg = C / groups, lane = threadIdx.x;
  const double mean = mr[(size_t)blockIdx.x * 2 + 0], rstd = mr[(size_t)blockIdx.x * 2 + 1];
  double m1 = 0.0, m2 = 0.0;
  for (int cc = lane; cc < cg; cc += 64) {
    const int c = g * cg + cc;
    double s1 = 0.0, s2 = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) {
      const float* p = part + (((size_t)b * nchunk + ch) * C + c) * 2;
      s1 += (double)p[0];
      s2 += (double)p[1];
    }
    const double t2 = rstd * (s2 - mean * s1);
    const double sp1 = ss ? (double)ss[b * ss_stride + c] + 1.0 : 1.0;
    const double k = (double)gamma[c] * sp1;
    m1 += k * s1;
    m2 += k * t2;
    pg[((size_t)b * 4 + 0) * C + c] = (float)(sp1 * t2);
    pg[((size_t)b * 4 + 1) * C + c] = (float)(sp1 * s1);
    pg[((size_t)b * 4 + 2) * C + c] = (float)((double)gamma[c] * t2 + (double)beta[c] * s1);
    pg[((size_t)b * 4 + 3) * C + c] = (float)s1;
    bcoef[((size_t)b * 3 + 0) * C + c] = (float)(rstd * k);
  }
  for (int off = 32; off; off >>= 1) {
    m1 += __shfl_xor(m1, off);
    m2 += __shfl_xor(m2, off);
  }
  const double n = (double)hw * (double)cg;
  m1 /= n;
  m2 /= n;
  const float Q = (float)(-rstd * rstd * m2), R = (float)(-rstd * m1 + rstd * rstd * m2 * mean);
  for (int cc = lane; cc < cg; cc += 64) {
    const int c = g * cg + cc;
    bcoef[((size_t)b * 3 + 1) * C + c] = Q;
    bcoef[((size_t)b * 3 + 2) * C + c] = R;
  }
}

// dy = P*dz + Q*y + R, dz = dout * SiLU'(a*y + c)
__global__ __launch_bounds__(256) void gn_silu_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                                const float* __restrict__ coef,
                                                                const float* __restrict__ bcoef, float* __restrict__ dy,
                                                                int64_t per_sample4, int C, int64_t total4) {
  const int C4 = C >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / per_sample4);
    const int c = (int)(i % C4) * 4;
    const float4 a = ld4(coef + (size_t)(b * 2 + 0) * C + c), cc = ld4(coef + (size_t)(b * 2 + 1) * C + c);
    const float4 P = ld4(bcoef + (size_t)(b * 3 + 0) * C + c), Q = ld4(bcoef + (size_t)(b * 3 + 1) * C + c);
    const float4 R = ld4(bcoef + (size_t)(b * 3 + 2) * C + c);
    const float4 d = ld4(dout + i * 4), v = ld4(y + i * 4);
    float4 o;
    o.x = fmaf(P.x, d.x * silu_grad(fmaf(a.x, v.x, cc.x)), fmaf(Q.x, v.x, R.x));
    o.y = fmaf(P.y, d.y * silu_grad(fmaf(a.y, v.y, cc.y)), fmaf(Q.y, v.y, R.y));
    o.z = fmaf(P.z, d.z * silu_grad(fmaf(a.z, v.z, cc.z)), fmaf(Q.z, v.z, R.z));
    o.w = fmaf(P.w, d.w * silu_grad(fmaf(a.w, v.w, cc.w)), fmaf(Q.w, v.w, R.w));
    st4(dy + i * 4, o);
  }
}

// out[i] = sum_b in[b][i]  (per-sample / per-block parameter-gradient parts -> the parameter gradient).  A workgroup owns
// 16 consecutive outputs and spreads b over 16 thread groups: group g adds rows g, g+16, ... in order, then the 16 sums
// are combined by a fixed tree — deterministic, and a long batch axis (1024 block partials) still fills the machine.
__global__ __launch_bounds__(256) void sum_over_batch_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                             int64_t per) {
  __shared__ float red[16][17];
  const int j = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + j;
  float s = 0.f;
  if (i < per) {
    const float* src = in + i;
    int b = g;
    for (; b + 48 < B; b += 64) {  // four loads in flight
      const float v0 = src[(size_t)b * per], v1 = src[(size_t)(b + 16) * per];
      const float v2 = src[(size_t)(b + 32) * per], v3 = src[(size_t)(b + 48) * per];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; b < B; b += 16) s += src[(size_t)b * per];
  }
  red[g][j] = s;
  __syncthreads();
  if (g == 0 && i < per) {
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = red[k][j];
#pragma unroll
    for (int w = 8; w; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; ++k) t[k] = t[2 * k] + t[2 * k + 1];
    out[i] = t[0];
  }
}

// weight standardisation backward, one workgroup per output channel: wh = (w - m)*r, r = rsqrt(var + eps):
//   dw = r * (dwh - mean(dwh) - wh * mean(dwh * wh))
__global__ __launch_bounds__(256) void ws_backward_kernel(const float* __restrict__ w, const float* __restrict__ dwh,
                                                          float* __restrict__ dw, int K, float eps) {
  __shared__ double red[4 * 4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const float* wr = w + (size_t)o * K;
  const float* gr = dwh + (size_t)o * K;
  auto block_sum = [&](double v, int slot) {
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) red[slot * 4 + (tid >> 6)] = v;
    __syncthreads();
    return red[slot * 4 + 0] + red[slot * 4 + 1] + red[slot * 4 + 2] + red[slot * 4 + 3];
  };
  double s = 0.0;
  for (int i = tid; i < K; i += 256) s += (double)wr[i];
  const double mean = block_sum(s, 0) / K;
  double q = 0.0;
  for (int i = tid; i < K; i += 256) {
    const double d = (double)wr[i] - mean;
    q += d * d;
  }
  const double r = 1.0 / sqrt(block_sum(q, 1) / K + (double)eps);
  double g1 = 0.0, g2 = 0.0;
  for (int i = tid; i < K; i += 256) {
    const double wh = ((double)wr[i] - mean) * r;
    g1 += (double)gr[i];
    g2 += (double)gr[i] * wh;
  }
  const double mg = block_sum(g1, 2) / K, mgw = block_sum(g2, 3) / K;
  for (int i = tid; i < K; i += 256) {
    const double wh = ((double)wr[i] - mean) * r;
    dw[(size_t)o * K + i] = (float)(r * ((double)gr[i] - mg - wh * mgw));
  }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int dmh_gn_bwd_chunks(int HW) { return cdiv(HW, GNB_PCH); }

// dout: gradient wrt SiLU(GN(y)) [B][HW][C]; y, coef, mr: what the forward pass saw / saved; ss as in dmh_gn_finalize.
// dy [B][HW][C]; pg [B][4][C]; part [B][dmh_gn_bwd_chunks(HW)][C][2] and bcoef [B][3][C] are work buffers.
extern "C" int dmh_gn_silu_backward(const float* dout, const float* y, const float* coef, const float* mr,
                                    const float* gamma, const float* beta, const float* ss, int64_t ss_stride, float* dy,
                                    float* pg, float* part, float* bcoef, int B, int HW, int C, int groups,
                                    void* stream) {
  DMH_REQUIRE(dout && y && coef && mr && gamma && beta && dy && pg && part && bcoef, "dmh_gn_silu_backward: null pointer");
  DMH_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0 && groups > 0 && C % groups == 0 && C <= 4096,
              "dmh_gn_silu_backward: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = cdiv(HW, GNB_PCH);
  hipLaunchKernelGGL(gn_silu_bwd_reduce_kernel, dim3(B * nchunk), dim3(256), 0, st, dout, y, coef, part, HW, C, nchunk);
  DMH_CHECK_LAUNCH("dmh_gn_silu_backward(reduce)");
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B * groups), dim3(64), 0, st, part, nchunk, mr, gamma, beta, ss,
                     ss_stride, bcoef, pg, C, groups, HW);
  DMH_CHECK_LAUNCH("dmh_gn_silu_backward(finalize)");
  const int64_t per_sample4 = (int64_t)HW * C / 4, total4 = per_sample4 * B;
  const unsigned grid = (unsigned)(cdiv64(total4, 256) < 8192 ? cdiv64(total4, 256) : 8192);
  hipLaunchKernelGGL(gn_silu_bwd_apply_kernel, dim3(grid), dim3(256), 0, st, dout, y, coef, bcoef, dy, per_sample4, C,
                     total4);
  DMH_CHECK_LAUNCH("dmh_gn_silu_backward(apply)");
  return DMH_OK;
}

extern "C" int dmh_sum_over_batch(const float* in, float* out, int B, int64_t per, void* stream) {
  DMH_REQUIRE(in && out && B > 0 && per > 0, "dmh_sum_over_batch: bad arguments");
  hipLaunchKernelGGL(sum_over_batch_kernel, dim3((unsigned)cdiv64(per, 16)), dim3(256), 0, (hipStream_t)stream, in, out,
                     B, per);
  DMH_CHECK_LAUNCH("dmh_sum_over_batch");
  return DMH_OK;
}

extern "C" int dmh_ws_backward(const float* w, const float* dwh, float* dw, int Cout, int K, float eps, void* stream) {
  DMH_REQUIRE(w && dwh && dw && Cout > 0 && K > 0, "dmh_ws_backward: bad arguments");
  hipLaunchKernelGGL(ws_backward_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, dwh, dw, K, eps);
  DMH_CHECK_LAUNCH("dmh_ws_backward");
  return DMH_OK;
}

// ------------------------------------------------------------------------------------------ channel LayerNorm backward
// N4 (CFG:137-141) out = (x - mean_c) * rstd * g [+ res]:  xh = (x - mean)*rstd, dxh = dout*g,
//   dx = rstd * (dxh - mean_c(dxh) - xh * mean_c(dxh*xh)),   dg[c] = sum_pixels dout*xh   (d res = dout)
// mean / rstd are recomputed from x (two-pass, as the forward).  LPP lanes share a pixel, NV float4 each; every
// workgroup also emits its partial dg row (fixed order inside; rows are added by dmh_sum_over_batch).
template <int LPP, int NV>
__global__ __launch_bounds__(256) void chan_layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ dout, float* __restrict__ dx,
                                                                 float* __restrict__ dg_part, int64_t npix, int C,
                                                                 float eps) {
  extern __shared__ float red[];  // [256 / LPP pixel groups][C] for the dg reduction
  const int C4 = C >> 2;
  const int sub = threadIdx.x % LPP, grp = threadIdx.x / LPP;
  const int64_t pix_per_block = 256 / LPP;
  float4 dgacc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) dgacc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t pix = (int64_t)blockIdx.x * pix_per_block + grp; pix < npix; pix += (int64_t)gridDim.x * pix_per_block) {
    float4 v[NV], d[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      d[j] = v[j];
      if (q < C4) {
        v[j] = ld4(x + pix * C + q * 4);
        d[j] = ld4(dout + pix * C + q * 4);
      }
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    s = lanes_sum<LPP>(s);
    const float mean = s / (float)C;
    float qsum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      if (q < C4) {
        const float ex = v[j].x - mean, ey = v[j].y - mean, ez = v[j].z - mean, ew = v[j].w - mean;
        qsum += (ex * ex + ey * ey) + (ez * ez + ew * ew);
      }
    }
    qsum = lanes_sum<LPP>(qsum);
    const float rstd = 1.0f / sqrtf(qsum / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
    float4 xh[NV], dxh[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      dxh[j] = xh[j];
      if (q < C4) {
        const float4 gg = ld4(g + q * 4);
        xh[j] = make_float4((v[j].x - mean) * rstd, (v[j].y - mean) * rstd, (v[j].z - mean) * rstd, (v[j].w - mean) * rstd);
        dxh[j] = make_float4(d[j].x * gg.x, d[j].y * gg.y, d[j].z * gg.z, d[j].w * gg.w);
        m1 += (dxh[j].x + dxh[j].y) + (dxh[j].z + dxh[j].w);
        m2 += (dxh[j].x * xh[j].x + dxh[j].y * xh[j].y) + (dxh[j].z * xh[j].z + dxh[j].w * xh[j].w);
        dgacc[j].x = fmaf(d[j].x, xh[j].x, dgacc[j].x);
        dgacc[j].y = fmaf(d[j].y, xh[j].y, dgacc[j].y);
        dgacc[j].z = fmaf(d[j].z, xh[j].z, dgacc[j].z);
        dgacc[j].w = fmaf(d[j].w, xh[j].w, dgacc[j].w);
      }
    }
    m1 = lanes_sum<LPP>(m1);
    m2 = lanes_sum<LPP>(m2);
    m1 /= (float)C;
    m2 /= (float)C;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      if (q < C4) {
        float4 o;
        o.x = rstd * (dxh[j].x - m1 - xh[j].x * m2);
        o.y = rstd * (dxh[j].y - m1 - xh[j].y * m2);
        o.z = rstd * (dxh[j].z - m1 - xh[j].z * m2);
        o.w = rstd * (dxh[j].w - m1 - xh[j].w * m2);
        st4(dx + pix * C + q * 4, o);
      }
    }
  }
  // dg of this workgroup: pixel groups added in a fixed order
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int q = sub + j * LPP;
    if (q < C4) st4(red + (size_t)grp * C + q * 4, dgacc[j]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int gI = 0; gI < (int)pix_per_block; ++gI) s += red[(size_t)gI * C + c];
    dg_part[(size_t)blockIdx.x * C + c] = s;
  }
}

#define DMH_LNB_BLOCKS 256  // partial dg rows

template <int LPP, int NV>
static int launch_lnb(const float* x, const float* g, const float* dout, float* dx, float* dg_part, int64_t npix, int C,
                      float eps, hipStream_t st) {
  const size_t lds = (size_t)(256 / LPP) * C * 4;
  hipLaunchKernelGGL((chan_layernorm_bwd_kernel<LPP, NV>), dim3(DMH_LNB_BLOCKS), dim3(256), lds, st, x, g, dout, dx,
                     dg_part, npix, C, eps);
  DMH_CHECK_LAUNCH("dmh_chan_layernorm_backward");
  return DMH_OK;
}

extern "C" int dmh_lnb_blocks(void) { return DMH_LNB_BLOCKS; }

// dx [npix][C]; dg_part [dmh_lnb_blocks()][C] partial rows of dg (add them with dmh_sum_over_batch)
extern "C" int dmh_chan_layernorm_backward(const float* x, const float* g, const float* dout, float* dx, float* dg_part,
                                           int64_t npix, int C, float eps, void* stream) {
  DMH_REQUIRE(x && g && dout && dx && dg_part, "dmh_chan_layernorm_backward: null pointer");
  DMH_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && C <= 1024, "dmh_chan_layernorm_backward: unsupported C=%d", C);
  hipStream_t st = (hipStream_t)stream;
  const int C4 = C / 4;
  if (C4 <= 8) return launch_lnb<8, 1>(x, g, dout, dx, dg_part, npix, C, eps, st);
  if (C4 <= 16) return launch_lnb<16, 1>(x, g, dout, dx, dg_part, npix, C, eps, st);
  if (C4 <= 32) return launch_lnb<32, 1>(x, g, dout, dx, dg_part, npix, C, eps, st);
  if (C4 <= 64) return launch_lnb<64, 1>(x, g, dout, dx, dg_part, npix, C, eps, st);
  if (C4 <= 128) return launch_lnb<64, 2>(x, g, dout, dx, dg_part, npix, C, eps, st);
  return launch_lnb<64, 4>(x, g, dout, dx, dg_part, npix, C, eps, st);
}
