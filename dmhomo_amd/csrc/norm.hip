// GroupNorm finalisation, GN+SiLU+residual, channel LayerNorm — the HBM-bound glue between
// the matrix-core kernels.  NHWC fp32, 16 B per lane per access.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// N2: reduce the per-tile (sum, sum^2) partials of dmh_conv2d for one (sample, group) in f64 and
// emit the affine the consumer applies:  y = a*x + b  ==  ((x-mean)*rstd*gamma+beta)*(scale+1)+shift
// One workgroup of four waves per (sample, group); fixed reduction order.  (Round 1 used one wave: at 128x128 its lanes
// walked 16 dependent rounds of partials, 6.7 us for a launch that happens 38 times per UNet pass.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ stats, int tiles,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ ss,
                                                          int64_t ss_stride, float* __restrict__ coef, int C, int groups,
                                                          int hw, float eps, float* __restrict__ mr,
                                                          float* __restrict__ bound, const int32_t* __restrict__ rows) {
  __shared__ double red[2][4];
  __shared__ float bred[4];
  const int jb = blockIdx.x / groups, g = blockIdx.x % groups;
  if (jb >= dmh_rows_n(rows, gridDim.x / groups)) return;   // (a row subset: inactive rows retire, common.h)
  const int b = dmh_rows_phys(rows, jb);
  const int bg = b * groups + g;
  const int cg = C / groups;
  const int lane = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  const int total = tiles * cg;
  for (int i = lane; i < total; i += 256) {
    const int tile = i / cg, cc = i % cg;
    const float2 st = *reinterpret_cast<const float2*>(stats + ((size_t)(b * tiles + tile) * C + g * cg + cc) * 2);
    s1 += (double)st.x;
    s2 += (double)st.y;
  }
  for (int off = 32; off; off >>= 1) {
    s1 += __shfl_xor(s1, off);
    s2 += __shfl_xor(s2, off);
  }
  if ((lane & 63) == 0) {
    red[0][lane >> 6] = s1;
    red[1][lane >> 6] = s2;
  }
  __syncthreads();
  s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double n = (double)hw * (double)cg;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  if (mr && lane == 0) {  // saved for the backward pass: (mean, rstd) of this (sample, group)
    mr[(size_t)bg * 2 + 0] = meanf;
    mr[(size_t)bg * 2 + 1] = rstd;
  }
  // bound (optional): an upper bound of |a*x + b| over every element x of this (sample, group), for a consumer that needs
  // the magnitude of its prologue's input without looking at the data (conv_f16x3.hip's block scale): |x - mean| <=
  // sqrt(sum (x - mean)^2) <= sqrt(sum x^2) — the accumulated s2, a sum of positive terms, no cancellation — so
  // |a*x + b| <= |a| * sqrt(s2) + |b + mean*a|.  (sqrt(n) standard deviations when the mean is small: far above the largest
  // element, which costs a block-scaled fp16 split range, not precision.)
  // When |mean| is more than 64 standard deviations the bound is that much looser than the data (and the variance itself is
  // what cancellation left of it): +inf then tells the consumer to look at the data instead.
  const float spread = (s2 > 4096.0 * n * (var + (double)eps)) ? INFINITY : (float)sqrt(s2 > 0.0 ? s2 : 0.0);
  float bmax = 0.f;
  for (int cc = lane; cc < cg; cc += 256) {
    const int c = g * cg + cc;
    float a = rstd * gamma[c];
    float bb = beta[c] - meanf * a;
    if (ss) {
      const float sc = ss[b * ss_stride + c] + 1.0f;
      const float sh = ss[b * ss_stride + C + c];
      a = a * sc;
      bb = fmaf(bb, sc, sh);
    }
    coef[(size_t)(b * 2 + 0) * C + c] = a;
    coef[(size_t)(b * 2 + 1) * C + c] = bb;
    // (spread == +inf: the whole group declines, whatever its gains are — 0 * inf would be a NaN that fmaxf drops)
    bmax = fmaxf(bmax, spread < INFINITY ? fmaf(fabsf(a), spread, fabsf(fmaf(meanf, a, bb))) : INFINITY);
  }
  if (bound) {
    for (int off = 32; off; off >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, off));
    if ((lane & 63) == 0) bred[lane >> 6] = bmax;
    __syncthreads();
    if (lane == 0) bound[bg] = fmaxf(fmaxf(bred[0], bred[1]), fmaxf(bred[2], bred[3]));
  }
}

// out = SiLU(a*y + b) + res, elementwise NHWC, a/b per (sample, channel)
// LPP > 0 (= C / 4 lanes per pixel): also the per-pixel (mean, rstd) of the channel LayerNorm of `out`, for the fused
// LinearAttention that follows a ResnetBlock (CFG:176-183 after :241) — the lanes of a pixel and the order of the sums are
// those of pixel_stats_kernel<LPP, 1>, so the values are bitwise the ones that kernel would compute from `out`, without
// reading it again.  Every lane of a wave runs every iteration (clamped index, guarded store): the lane sums need them all.
template <int LPP>
__global__ __launch_bounds__(256) void gn_silu_residual_kernel(const float* __restrict__ y,
                                                               const float* __restrict__ coef,
                                                               const float* __restrict__ res, float* __restrict__ out,
                                                               int64_t per_sample4, int C, int64_t total4_all,
                                                               float* __restrict__ pstats, float eps,
                                                               const int32_t* __restrict__ rows) {
  const int C4 = C >> 2;
  const int lane = threadIdx.x & 63;
  // (a row subset, common.h: the flat index walks the active rows; tensors are addressed by physical row)
  const int64_t total4 = rows ? per_sample4 * rows[0] : total4_all;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 - lane < total4; i0 += (int64_t)gridDim.x * 256) {
    const bool ok = i0 < total4;
    if (LPP == 0 && !ok) break;
    const int64_t il = ok ? i0 : total4 - 1;
    const int jb = (int)(il / per_sample4);
    const int b = dmh_rows_phys(rows, jb);
    const int64_t i = il + (int64_t)(b - jb) * per_sample4;
    const int c = (int)(i % C4) * 4;
    const float4 a = ld4(coef + (size_t)(b * 2 + 0) * C + c);
    const float4 bb = ld4(coef + (size_t)(b * 2 + 1) * C + c);
    const float4 v = ld4(y + i * 4);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) r = ld4(res + i * 4);
    float4 o;
    o.x = silu_fast(fmaf(a.x, v.x, bb.x)) + r.x;
    o.y = silu_fast(fmaf(a.y, v.y, bb.y)) + r.y;
    o.z = silu_fast(fmaf(a.z, v.z, bb.z)) + r.z;
    o.w = silu_fast(fmaf(a.w, v.w, bb.w)) + r.w;
    if (ok) st4(out + i * 4, o);
    if (LPP > 0) {
      float s = 0.f;
      s += (o.x + o.y) + (o.z + o.w);
      s = lanes_sum<(LPP > 0 ? LPP : 1)>(s);
      const float mean = s / (float)C;
      float qsum = 0.f;
      const float dx = o.x - mean, dy = o.y - mean, dz = o.z - mean, dw = o.w - mean;
      qsum += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      qsum = lanes_sum<(LPP > 0 ? LPP : 1)>(qsum);
      if (ok && (threadIdx.x % (LPP > 0 ? LPP : 1)) == 0) {
        const int64_t pix = i / C4;
        pstats[pix * 2 + 0] = mean;
        pstats[pix * 2 + 1] = 1.0f / sqrtf(qsum / (float)C + eps);
      }
    }
  }
}

// N4: per-pixel LayerNorm over the (contiguous) channel axis. LPP lanes share a pixel, each
// holding NV float4; two-pass (mean, then centred second moment) in registers.
template <int LPP, int NV>
__global__ __launch_bounds__(256) void chan_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ res, float* __restrict__ out,
                                                             int64_t npix_all, int C, float eps,
                                                             const int32_t* __restrict__ rows, int64_t pix_per_row) {
  const int C4 = C >> 2;
  const int sub = threadIdx.x % LPP;
  const int64_t pix_per_block = 256 / LPP;
  const int64_t npix = rows ? pix_per_row * rows[0] : npix_all;   // (a row subset, common.h)
  for (int64_t pl = (int64_t)blockIdx.x * pix_per_block + threadIdx.x / LPP; pl < npix;
       pl += (int64_t)gridDim.x * pix_per_block) {
    const int64_t pix = rows ? (int64_t)rows[1 + pl / pix_per_row] * pix_per_row + pl % pix_per_row : pl;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < C4) v[j] = ld4(x + pix * C + q * 4);
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    s = lanes_sum<LPP>(s);
    const float mean = s / (float)C;
    float qsum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      if (q < C4) {
        const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
        qsum += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    qsum = lanes_sum<LPP>(qsum);
    const float rstd = 1.0f / sqrtf(qsum / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      if (q < C4) {
        const float4 gg = ld4(g + q * 4);
        float4 o;
        o.x = (v[j].x - mean) * rstd * gg.x;
        o.y = (v[j].y - mean) * rstd * gg.y;
        o.z = (v[j].z - mean) * rstd * gg.z;
        o.w = (v[j].w - mean) * rstd * gg.w;
        if (res) {
          const float4 r = ld4(res + pix * C + q * 4);
          o.x += r.x;
          o.y += r.y;
          o.z += r.z;
          o.w += r.w;
        }
        st4(out + pix * C + q * 4, o);
      }
    }
  }
}

extern "C" int dmh_gn_finalize(const float* stats, int tiles, const float* gamma, const float* beta, const float* ss,
                               int64_t ss_stride, float* coef, int B, int C, int groups, int hw, float eps,
                               const int32_t* rows, void* stream) {
  DMH_REQUIRE(stats && gamma && beta && coef, "dmh_gn_finalize: null pointer");
  DMH_REQUIRE(B > 0 && C > 0 && groups > 0 && C % groups == 0 && tiles > 0 && hw > 0, "dmh_gn_finalize: bad shape");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, stats, tiles, gamma, beta,
                     ss, ss_stride, coef, C, groups, hw, eps, (float*)nullptr, (float*)nullptr, rows);
  DMH_CHECK_LAUNCH("dmh_gn_finalize");
  return DMH_OK;
}

// same, also writing bound [B][groups]: an upper bound of |a*x + b| over each (sample, group) of the tensor the statistics
// were taken from — DmhConv.in_bound of the conv that applies this affine in its prologue
extern "C" int dmh_gn_finalize_bound(const float* stats, int tiles, const float* gamma, const float* beta, const float* ss,
                                     int64_t ss_stride, float* coef, float* bound, int B, int C, int groups, int hw,
                                     float eps, const int32_t* rows, void* stream) {
  DMH_REQUIRE(stats && gamma && beta && coef && bound, "dmh_gn_finalize_bound: null pointer");
  DMH_REQUIRE(B > 0 && C > 0 && groups > 0 && C % groups == 0 && tiles > 0 && hw > 0, "dmh_gn_finalize_bound: bad shape");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, stats, tiles, gamma, beta,
                     ss, ss_stride, coef, C, groups, hw, eps, (float*)nullptr, bound, rows);
  DMH_CHECK_LAUNCH("dmh_gn_finalize_bound");
  return DMH_OK;
}

// same, also saving (mean, rstd) per (sample, group) for the backward pass: mr [B][groups][2]
extern "C" int dmh_gn_finalize_train(const float* stats, int tiles, const float* gamma, const float* beta,
                                     const float* ss, int64_t ss_stride, float* coef, float* mr, int B, int C,
                                     int groups, int hw, float eps, void* stream) {
  DMH_REQUIRE(stats && gamma && beta && coef && mr, "dmh_gn_finalize_train: null pointer");
  DMH_REQUIRE(B > 0 && C > 0 && groups > 0 && C % groups == 0 && tiles > 0 && hw > 0, "dmh_gn_finalize_train: bad shape");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, stats, tiles, gamma, beta,
                     ss, ss_stride, coef, C, groups, hw, eps, mr, (float*)nullptr, (const int32_t*)nullptr);
  DMH_CHECK_LAUNCH("dmh_gn_finalize_train");
  return DMH_OK;
}

extern "C" int dmh_gn_silu_residual(const float* y, const float* coef, const float* res, float* out, int B, int HW,
                                    int C, const int32_t* rows, void* stream) {
  DMH_REQUIRE(y && coef && out, "dmh_gn_silu_residual: null pointer");
  DMH_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 4 == 0, "dmh_gn_silu_residual: bad shape (C %% 4 != 0?)");
  const int64_t per_sample4 = (int64_t)HW * C / 4;
  const int64_t total4 = per_sample4 * B;
  const unsigned grid = (unsigned)(cdiv64(total4, 256) < 8192 ? cdiv64(total4, 256) : 8192);
  hipLaunchKernelGGL(gn_silu_residual_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, coef, res, out,
                     per_sample4, C, total4, (float*)nullptr, 0.f, rows);
  DMH_CHECK_LAUNCH("dmh_gn_silu_residual");
  return DMH_OK;
}

// same + pstats [B*HW][2] = (mean, rstd) over the channels of every output pixel, as dmh_pixel_stats(out) would give
extern "C" int dmh_gn_silu_residual_stats(const float* y, const float* coef, const float* res, float* out, float* pstats,
                                          int B, int HW, int C, float eps, const int32_t* rows, void* stream) {
  DMH_REQUIRE(y && coef && out && pstats, "dmh_gn_silu_residual_stats: null pointer");
  DMH_REQUIRE(B > 0 && HW > 0 && (C == 64 || C == 128 || C == 256),
              "dmh_gn_silu_residual_stats: C must be 64, 128 or 256 (got %d)", C);
  const int64_t per_sample4 = (int64_t)HW * C / 4;
  const int64_t total4 = per_sample4 * B;
  const unsigned grid = (unsigned)(cdiv64(total4, 256) < 8192 ? cdiv64(total4, 256) : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (C == 64)
    hipLaunchKernelGGL(gn_silu_residual_kernel<16>, dim3(grid), dim3(256), 0, st, y, coef, res, out, per_sample4, C, total4,
                       pstats, eps, rows);
  else if (C == 128)
    hipLaunchKernelGGL(gn_silu_residual_kernel<32>, dim3(grid), dim3(256), 0, st, y, coef, res, out, per_sample4, C, total4,
                       pstats, eps, rows);
  else
    hipLaunchKernelGGL(gn_silu_residual_kernel<64>, dim3(grid), dim3(256), 0, st, y, coef, res, out, per_sample4, C, total4,
                       pstats, eps, rows);
  DMH_CHECK_LAUNCH("dmh_gn_silu_residual_stats");
  return DMH_OK;
}

template <int LPP, int NV>
static int launch_ln(const float* x, const float* g, const float* res, float* out, int64_t npix, int C, float eps,
                     const int32_t* rows, int64_t ppr, hipStream_t st) {
  const int64_t ppb = 256 / LPP;
  const int64_t need = cdiv64(npix, ppb);
  const unsigned grid = (unsigned)(need < 16384 ? need : 16384);
  hipLaunchKernelGGL((chan_layernorm_kernel<LPP, NV>), dim3(grid), dim3(256), 0, st, x, g, res, out, npix, C, eps, rows, ppr);
  DMH_CHECK_LAUNCH("dmh_chan_layernorm");
  return DMH_OK;
}

// per-pixel (mean, rstd) of the channel LayerNorm, for consumers that apply it while staging (linattn_fused.hip);
// same two-pass arithmetic as chan_layernorm_kernel
template <int LPP, int NV>
__global__ __launch_bounds__(256) void pixel_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                          int64_t npix_all, int C, float eps,
                                                          const int32_t* __restrict__ rows, int64_t pix_per_row) {
  const int C4 = C >> 2;
  const int sub = threadIdx.x % LPP;
  const int64_t pix_per_block = 256 / LPP;
  const int64_t npix = rows ? pix_per_row * rows[0] : npix_all;   // (a row subset, common.h)
  for (int64_t pl = (int64_t)blockIdx.x * pix_per_block + threadIdx.x / LPP; pl < npix;
       pl += (int64_t)gridDim.x * pix_per_block) {
    const int64_t pix = rows ? (int64_t)rows[1 + pl / pix_per_row] * pix_per_row + pl % pix_per_row : pl;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < C4) v[j] = ld4(x + pix * C + q * 4);
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    s = lanes_sum<LPP>(s);
    const float mean = s / (float)C;
    float qsum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int q = sub + j * LPP;
      if (q < C4) {
        const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
        qsum += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    qsum = lanes_sum<LPP>(qsum);
    if (sub == 0) {
      stats[pix * 2 + 0] = mean;
      stats[pix * 2 + 1] = 1.0f / sqrtf(qsum / (float)C + eps);
    }
  }
}

template <int LPP, int NV>
static int launch_ps(const float* x, float* stats, int64_t npix, int C, float eps, const int32_t* rows, int64_t ppr,
                     hipStream_t st) {
  const int64_t ppb = 256 / LPP;
  const int64_t need = cdiv64(npix, ppb);
  const unsigned grid = (unsigned)(need < 16384 ? need : 16384);
  hipLaunchKernelGGL((pixel_stats_kernel<LPP, NV>), dim3(grid), dim3(256), 0, st, x, stats, npix, C, eps, rows, ppr);
  DMH_CHECK_LAUNCH("dmh_pixel_stats");
  return DMH_OK;
}

extern "C" int dmh_pixel_stats(const float* x, float* stats, int64_t npix, int C, float eps, const int32_t* rows,
                               int64_t pix_per_row, void* stream) {
  DMH_REQUIRE(x && stats, "dmh_pixel_stats: null pointer");
  DMH_REQUIRE(!rows || (pix_per_row > 0 && npix % pix_per_row == 0), "dmh_pixel_stats: rows needs pix_per_row dividing npix");
  DMH_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && C <= 2048, "dmh_pixel_stats: unsupported C=%d", C);
  hipStream_t st = (hipStream_t)stream;
  const int C4 = C / 4;
  if (C4 <= 8) return launch_ps<8, 1>(x, stats, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 16) return launch_ps<16, 1>(x, stats, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 32) return launch_ps<32, 1>(x, stats, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 64) return launch_ps<64, 1>(x, stats, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 128) return launch_ps<64, 2>(x, stats, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 256) return launch_ps<64, 4>(x, stats, npix, C, eps, rows, pix_per_row, st);
  return launch_ps<64, 8>(x, stats, npix, C, eps, rows, pix_per_row, st);
}

extern "C" int dmh_chan_layernorm(const float* x, const float* g, const float* res, float* out, int64_t npix, int C,
                                  float eps, const int32_t* rows, int64_t pix_per_row, void* stream) {
  DMH_REQUIRE(x && g && out, "dmh_chan_layernorm: null pointer");
  DMH_REQUIRE(!rows || (pix_per_row > 0 && npix % pix_per_row == 0), "dmh_chan_layernorm: rows needs pix_per_row dividing npix");
  DMH_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && C <= 2048, "dmh_chan_layernorm: unsupported C=%d", C);
  hipStream_t st = (hipStream_t)stream;
  const int C4 = C / 4;
  if (C4 <= 2) return launch_ln<2, 1>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 4) return launch_ln<4, 1>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 8) return launch_ln<8, 1>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 16) return launch_ln<16, 1>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 32) return launch_ln<32, 1>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 64) return launch_ln<64, 1>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 128) return launch_ln<64, 2>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  if (C4 <= 256) return launch_ln<64, 4>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
  return launch_ln<64, 8>(x, g, res, out, npix, C, eps, rows, pix_per_row, st);
}
