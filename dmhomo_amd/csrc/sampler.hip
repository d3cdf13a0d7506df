// K6 — the elementwise glue of the sampling loop at the NCHW API boundary: network input
// assembly, final 1x1 conv back to NCHW, classifier-free-guidance blend + x0 clamp + DDIM /
// DDPM update, uint8 export.  Mirrors the reference's fp32 op order (no FMA contraction).
#include "common.h"

#pragma clang fp contract(off)

// out NHWC [reps*B][HW][Cpad] <- cat(a, b*m) zero padded; sample r reads source r % B
__global__ __launch_bounds__(256) void assemble_input_kernel(const float* __restrict__ a, int Ca,
                                                             const float* __restrict__ b, int Cb,
                                                             const float* __restrict__ m, float* __restrict__ out,
                                                             int B, int HW, int Cpad, int64_t total) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int c = (int)(idx % Cpad);
    const int64_t rp = idx / Cpad;
    const int p = (int)(rp % HW);
    const int r = (int)(rp / HW);
    const int s = r % B;
    float v = 0.f;
    if (c < Ca) {
      v = a[((size_t)s * Ca + c) * HW + p];
    } else if (c < Ca + Cb) {
      v = b[((size_t)s * Cb + (c - Ca)) * HW + p];
      if (m) v = v * m[(size_t)s * HW + p];
    }
    out[idx] = v;
  }
}

// the same for Cpad <= 16 (every UNet input of the path: 12, 8, 4): one thread per (row, pixel) — the NCHW reads of a channel
// are consecutive over the lanes and a thread writes its pixel's Cpad floats as 16 B pieces (the element-per-thread form above
// read with a stride of HW floats between neighbouring lanes: 20.8 us for the 20 MB of the headline step, on the serial path)
template <int CP4>
__global__ __launch_bounds__(256) void assemble_input_pix_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b,
                                                                 int Cb, const float* __restrict__ m, float* __restrict__ out,
                                                                 int B, int HW, int64_t npix) {
  for (int64_t rp = (int64_t)blockIdx.x * 256 + threadIdx.x; rp < npix; rp += (int64_t)gridDim.x * 256) {
    const int p = (int)(rp % HW);
    const int s = (int)(rp / HW) % B;
    float v[CP4 * 4];
#pragma unroll
    for (int c = 0; c < CP4 * 4; ++c) {
      float x = 0.f;
      if (c < Ca) x = a[((size_t)s * Ca + c) * HW + p];
      else if (c < Ca + Cb) x = b[((size_t)s * Cb + (c - Ca)) * HW + p];
      v[c] = x;
    }
    if (m) {
      const float mv = m[(size_t)s * HW + p];
#pragma unroll
      for (int c = 0; c < CP4 * 4; ++c)
        if (c >= Ca && c < Ca + Cb) v[c] = v[c] * mv;
    }
#pragma unroll
    for (int q = 0; q < CP4; ++q) st4(out + rp * (CP4 * 4) + q * 4, make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]));
  }
}

// final_conv: 16 lanes per pixel, each a float4 slice of the channels; Cout <= 16 dot products
// reduced over the 16 lanes with shuffles, written NCHW.
template <int COUT>
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int64_t npix, int HW, int C) {
  const int sub = threadIdx.x & 15;
  const int C4 = C >> 2;
  // C <= 64 (the DGM UNet): a lane owns ONE channel quad, so its COUT weight quads are loaded once, not once per pixel
  const bool one = C4 <= 16;
  float4 wq[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) wq[o] = (one && sub < C4) ? ld4(w + (size_t)o * C + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float bo = 0.f;   // lane o of a pixel's 16 writes output channel o
#pragma unroll
  for (int o = 0; o < COUT; ++o)
    if (sub == o && bias) bo = bias[o];
  // U pixels per thread group and iteration: their loads are in flight together (round 2: one pixel per iteration left a
  // single 16 B load per lane between dependent reduction chains)
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t pix0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); pix0 < npix; pix0 += stride * U) {
    float4 v[U];
    if (one) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t pix = pix0 + u * stride;
        v[u] = (sub < C4 && pix < npix) ? ld4(x + pix * C + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pix = pix0 + u * stride;
      if (pix >= npix) break;   // (uniform over the 16 lanes of a pixel; the lane sums below stay inside them)
      float acc[COUT];
#pragma unroll
      for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
      if (one) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          acc[o] = fmaf(v[u].x, wq[o].x, acc[o]);
          acc[o] = fmaf(v[u].y, wq[o].y, acc[o]);
          acc[o] = fmaf(v[u].z, wq[o].z, acc[o]);
          acc[o] = fmaf(v[u].w, wq[o].w, acc[o]);
        }
      } else {
        for (int q = sub; q < C4; q += 16) {
          const float4 vv = ld4(x + pix * C + q * 4);
#pragma unroll
          for (int o = 0; o < COUT; ++o) {
            const float4 ww = ld4(w + (size_t)o * C + q * 4);
            acc[o] = fmaf(vv.x, ww.x, acc[o]);
            acc[o] = fmaf(vv.y, ww.y, acc[o]);
            acc[o] = fmaf(vv.z, ww.z, acc[o]);
            acc[o] = fmaf(vv.w, ww.w, acc[o]);
          }
        }
      }
      float mine = 0.f;
#pragma unroll
      for (int o = 0; o < COUT; ++o) {
        const float t = row16_sum(acc[o]);   // (every lane of the pixel gets the sum)
        mine = sub == o ? t : mine;
      }
      if (sub < COUT) {
        const unsigned r = (unsigned)pix / (unsigned)HW, p = (unsigned)pix - r * (unsigned)HW;   // (npix < 2^31: checked at launch)
        out[((size_t)r * COUT + sub) * HW + p] = mine + bo;
      }
    }
  }
}

// torch.clamp(x, -1., 1.) (CFG:612,634): NaN stays NaN (fminf / fmaxf alone would turn it into -1 and hide a broken row)
__device__ __forceinline__ float clamp_pm1(float x) { return x != x ? x : fminf(fmaxf(x, -1.f), 1.f); }

__global__ __launch_bounds__(256) void sampler_step_kernel(DmhStep s, const float* __restrict__ mc,
                                                           const float* __restrict__ mn, const float* __restrict__ x,
                                                           const float* __restrict__ noise, float* __restrict__ img_out,
                                                           float* __restrict__ x_start, float* __restrict__ pred_noise,
                                                           int64_t n, const uint8_t* __restrict__ keep, int64_t per_row) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float mo;
    if (mn) {
      // keep (with model_null): row i / per_row of model_cond was only computed where keep != 0 — a row whose class the
      // conditional pass dropped (CFG:415-425) has the null pass's inputs, so its logits ARE the null logits
      const float nl = mn[i];
      mo = (keep && !keep[i / per_row]) ? nl : mc[i];
      mo = nl + (mo - nl) * s.cond_scale;  // CFG:410
    } else {
      mo = mc[i];
    }
    const float xt = x[i];
    float x0, pn;
    if (s.objective == 0) {  // pred_noise, CFG:614-617
      pn = mo;
      x0 = s.sqrt_recip_ac * xt - s.sqrt_recipm1_ac * pn;
      if (s.clip) x0 = clamp_pm1(x0);
    } else if (s.objective == 1) {  // pred_x0, CFG:619-622
      x0 = mo;
      if (s.clip) x0 = clamp_pm1(x0);
      pn = (s.sqrt_recip_ac * xt - x0) / s.sqrt_recipm1_ac;
    } else {  // pred_v, CFG:624-628
      x0 = s.sqrt_ac * xt - s.sqrt_1m_ac * mo;
      if (s.clip) x0 = clamp_pm1(x0);
      pn = (s.sqrt_recip_ac * xt - x0) / s.sqrt_recipm1_ac;
    }
    float o;
    if (s.mode == 0) {  // DDIM, CFG:705-707
      o = x0 * s.c0 + s.c1 * pn + s.c2 * noise[i];
    } else if (s.mode == 1) {  // last DDIM step, CFG:693-695
      o = x0;
    } else {  // DDPM posterior step, DDP:604-611,660: mean + exp(.5 logvar) * noise (noise == NULL at t == 0)
      o = s.c0 * x0 + s.c1 * xt;
      if (noise) o = o + s.c2 * noise[i];
    }
    img_out[i] = o;
    if (x_start) x_start[i] = x0;
    if (pred_noise) pred_noise[i] = pn;
  }
}

// the step kernel with its DmhStep read from device memory (dmh_sampler_step_dev); no __restrict__ on x / img_out: the
// replayed loop updates img in place (every element is read by the thread that writes it)
__global__ __launch_bounds__(256) void sampler_step_dev_kernel(const DmhStep* __restrict__ sp, const float* __restrict__ mc,
                                                               const float* __restrict__ mn, const float* x,
                                                               const float* __restrict__ noise, float* img_out,
                                                               float* __restrict__ x_start, float* __restrict__ pred_noise,
                                                               int64_t n, const uint8_t* __restrict__ keep, int64_t per_row) {
  const DmhStep s = *sp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float mo;
    if (mn) {
      const float nl = mn[i];
      mo = (keep && !keep[i / per_row]) ? nl : mc[i];   // (as sampler_step_kernel)
      mo = nl + (mo - nl) * s.cond_scale;  // CFG:410
    } else {
      mo = mc[i];
    }
    const float xt = x[i];
    float x0, pn;
    if (s.objective == 0) {
      pn = mo;
      x0 = s.sqrt_recip_ac * xt - s.sqrt_recipm1_ac * pn;
      if (s.clip) x0 = clamp_pm1(x0);
    } else if (s.objective == 1) {
      x0 = mo;
      if (s.clip) x0 = clamp_pm1(x0);
      pn = (s.sqrt_recip_ac * xt - x0) / s.sqrt_recipm1_ac;
    } else {
      x0 = s.sqrt_ac * xt - s.sqrt_1m_ac * mo;
      if (s.clip) x0 = clamp_pm1(x0);
      pn = (s.sqrt_recip_ac * xt - x0) / s.sqrt_recipm1_ac;
    }
    float o;
    if (s.mode == 0) {
      // a DDIM entry always comes with noise; the device-resident entry cannot be checked at launch, so a mis-sequenced
      // cursor (the last-step graph replayed on a non-last entry) shows up as NaN instead of a plausible wrong sample
      o = x0 * s.c0 + s.c1 * pn + s.c2 * (noise ? noise[i] : __builtin_nanf(""));
    } else if (s.mode == 1) {
      o = x0;
    } else {
      o = s.c0 * x0 + s.c1 * xt;
      if (noise) o = o + s.c2 * noise[i];
    }
    img_out[i] = o;
    if (x_start) x_start[i] = x0;
    if (pred_noise) pred_noise[i] = pn;
  }
}

__global__ __launch_bounds__(64) void sampler_seek_kernel(int32_t* cursor, int k, const DmhStep* __restrict__ table,
                                                          const int64_t* __restrict__ times, int S, DmhStep* cur,
                                                          int64_t* tcond, int B) {
  int c = k >= 0 ? k : *cursor + 1;
  c = c < S ? c : S - 1;
  __syncthreads();   // every lane has read the old cursor
  if (threadIdx.x == 0) {
    *cursor = c;
    *cur = table[c];
  }
  const int64_t t = times[c];
  for (int i = threadIdx.x; i < B; i += 64) tcond[i] = t;
}

__global__ __launch_bounds__(256) void affine_tail_kernel(float* __restrict__ x, int C, int HW, int c0, float scale,
                                                          float shift, int64_t total) {
  const int64_t per = (int64_t)(C - c0) * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per, r = i % per;
    float* p = x + (b * C + c0) * HW + r;
    *p = *p * scale + shift;
  }
}

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                                       float* __restrict__ out, int64_t per_sample, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / per_sample);
    out[i] = ca[b] * x0[i] + cb[b] * noise[i];
  }
}

// out = ca[b] * x (+ cb[b] * y) (/ dv[b]) (clamped to [-1, 1]): the per-sample affine combinations of CFG:586-608 with a
// timestep per row, in the reference's op order (two rounded products, one add, one divide)
__global__ __launch_bounds__(256) void rows_lincomb_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ dv, float* __restrict__ out,
                                                           int64_t per_sample, int64_t total, int clamp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / per_sample);
    float v = ca[b] * x[i];
    if (y) v = v + cb[b] * y[i];
    if (dv) v = v / dv[b];
    if (clamp) v = clamp_pm1(v);
    out[i] = v;
  }
}

__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ x, float* __restrict__ y, float scale,
                                                     float shift, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = x[i] * scale + shift;
}

__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out,
                                                       int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = img[i] * 255.f;
    out[i] = (uint8_t)(int)v;  // numpy astype(uint8) of a float in [0, 255]: truncation
  }
}

// per-sample mean of m*|a-b| (or m*(a-b)^2): 64 blocks per sample accumulate in f64, fixed order
__global__ __launch_bounds__(256) void diff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ m, int squared,
                                                           double* __restrict__ ws, int C, int HW) {
  __shared__ double red[4];
  const int s = blockIdx.y, blk = blockIdx.x;
  const int64_t per = (int64_t)C * HW;
  double acc = 0.0;
  for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < per; i += 64 * 256) {
    const float d = a[s * per + i] - b[s * per + i];
    float v = squared ? d * d : fabsf(d);
    if (m) v = m[(int64_t)s * HW + i % HW] * v;
    acc += (double)v;
  }
  for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) ws[s * 64 + blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void diff_final_kernel(const double* __restrict__ ws, float* __restrict__ out, int B, double inv_count) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B) return;
  double t = 0.0;
  for (int i = 0; i < 64; ++i) t += ws[s * 64 + i];
  out[s] = (float)(t * inv_count);
}

__global__ void loss_combine_kernel(const float* __restrict__ l, const float* __restrict__ ph,
                                    const float* __restrict__ w, float* __restrict__ out, int B) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double a = 0.0, p = 0.0;
  for (int i = 0; i < B; ++i) {
    a += (double)l[i];
    p += (double)(1.0f * w[i] * ph[i]);
  }
  out[0] = (float)(a / B) + (float)(p / B);
}

static unsigned grid_for(int64_t n, int per_block = 256) {
  const int64_t g = cdiv64(n, per_block);
  return (unsigned)(g < 16384 ? (g > 0 ? g : 1) : 16384);
}

extern "C" int dmh_assemble_input(const float* a, int Ca, const float* b, int Cb, const float* m, float* out, int B,
                                  int reps, int HW, int Cpad, void* stream) {
  DMH_REQUIRE(a && out && B > 0 && reps > 0 && HW > 0 && Ca > 0 && Cb >= 0, "dmh_assemble_input: bad arguments");
  DMH_REQUIRE(Cpad % 4 == 0 && Cpad >= Ca + Cb, "dmh_assemble_input: Cpad=%d must be a multiple of 4 >= %d", Cpad,
              Ca + Cb);
  DMH_REQUIRE(Cb == 0 || b, "dmh_assemble_input: b is NULL with Cb > 0");
  const int64_t total = (int64_t)reps * B * HW * Cpad;
  const int64_t npix = (int64_t)reps * B * HW;
  hipStream_t st = (hipStream_t)stream;
#define DMH_ASM(N)                                                                                                       \
  case N:                                                                                                                \
    hipLaunchKernelGGL((assemble_input_pix_kernel<N>), dim3(grid_for(npix)), dim3(256), 0, st, a, Ca, b, Cb, m, out, B, HW, npix); \
    break;
  switch (Cpad / 4) {
    DMH_ASM(1) DMH_ASM(2) DMH_ASM(3) DMH_ASM(4)
    default:
      hipLaunchKernelGGL(assemble_input_kernel, dim3(grid_for(total)), dim3(256), 0, st, a, Ca, b, Cb, m, out, B, HW, Cpad, total);
  }
#undef DMH_ASM
  DMH_CHECK_LAUNCH("dmh_assemble_input");
  return DMH_OK;
}

extern "C" int dmh_final_conv_nchw(const float* x, const float* w, const float* bias, float* out, int R, int HW, int C,
                                   int Cout, void* stream) {
  DMH_REQUIRE(x && w && out && R > 0 && HW > 0 && C > 0 && C % 4 == 0, "dmh_final_conv_nchw: bad arguments");
  const int64_t npix = (int64_t)R * HW;
  DMH_REQUIRE(npix < ((int64_t)1 << 31), "dmh_final_conv_nchw: %lld pixels (limit 2^31)", (long long)npix);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(npix, 64));   // four pixels per thread group and iteration
#define DMH_FC(N)                                                                                             \
  case N:                                                                                                     \
    hipLaunchKernelGGL((final_conv_kernel<N>), grid, dim3(256), 0, st, x, w, bias, out, npix, HW, C); \
    break;
  switch (Cout) {
    DMH_FC(1) DMH_FC(2) DMH_FC(3) DMH_FC(4) DMH_FC(6) DMH_FC(8) DMH_FC(12) DMH_FC(16)
    default:
      dmh_set_error("dmh_final_conv_nchw: unsupported Cout=%d (1,2,3,4,6,8,12,16)", Cout);
      return DMH_EINVAL;
  }
#undef DMH_FC
  DMH_CHECK_LAUNCH("dmh_final_conv_nchw");
  return DMH_OK;
}

extern "C" int dmh_sampler_step(const DmhStep* s, const float* model_cond, const float* model_null, const float* x,
                                const float* noise, float* img_out, float* x_start, float* pred_noise, int64_t n,
                                const uint8_t* keep, int64_t per_row, void* stream) {
  DMH_REQUIRE(s && model_cond && x && img_out && n > 0, "dmh_sampler_step: bad arguments");
  DMH_REQUIRE(!keep || (model_null && per_row > 0 && n % per_row == 0), "dmh_sampler_step: keep needs model_null and per_row dividing n");
  DMH_REQUIRE(s->objective >= 0 && s->objective <= 2 && s->mode >= 0 && s->mode <= 2, "dmh_sampler_step: bad enum");
  DMH_REQUIRE(s->mode != 0 || noise, "dmh_sampler_step: DDIM update needs noise");
  hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, *s, model_cond,
                     model_null, x, noise, img_out, x_start, pred_noise, n, keep, per_row);
  DMH_CHECK_LAUNCH("dmh_sampler_step");
  return DMH_OK;
}

extern "C" int dmh_sampler_step_dev(const DmhStep* cur_dev, const float* model_cond, const float* model_null, const float* x,
                                    const float* noise, float* img_out, float* x_start, float* pred_noise, int64_t n,
                                    const uint8_t* keep, int64_t per_row, void* stream) {
  DMH_REQUIRE(cur_dev && model_cond && x && img_out && n > 0, "dmh_sampler_step_dev: bad arguments");
  DMH_REQUIRE(!keep || (model_null && per_row > 0 && n % per_row == 0), "dmh_sampler_step_dev: keep needs model_null and per_row dividing n");
  hipLaunchKernelGGL(sampler_step_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cur_dev, model_cond,
                     model_null, x, noise, img_out, x_start, pred_noise, n, keep, per_row);
  DMH_CHECK_LAUNCH("dmh_sampler_step_dev");
  return DMH_OK;
}

// rows[0] = n, rows[1 ..] = the rows b < B with keep[b] != 0 in ascending order, then B .. B + extra - 1: the active-row
// subset of a classifier-free-guidance pass (common.h: dmh_rows_n / dmh_rows_phys).  One wave.
__global__ __launch_bounds__(64) void rows_from_keep_kernel(const uint8_t* __restrict__ keep, int B, int extra,
                                                            int32_t* __restrict__ rows) {
  int n = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + (int)threadIdx.x;
    const bool k = b < B && keep[b] != 0;
    const unsigned long long m = __ballot(k);
    if (k) rows[1 + n + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = b;
    n += __popcll(m);
  }
  for (int j = threadIdx.x; j < extra; j += 64) rows[1 + n + j] = B + j;
  if (threadIdx.x == 0) rows[0] = n + extra;
}

extern "C" int dmh_rows_from_keep(const uint8_t* keep, int B, int extra, int32_t* rows, void* stream) {
  DMH_REQUIRE(keep && rows && B > 0 && extra >= 0, "dmh_rows_from_keep: bad arguments");
  hipLaunchKernelGGL(rows_from_keep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, keep, B, extra, rows);
  DMH_CHECK_LAUNCH("dmh_rows_from_keep");
  return DMH_OK;
}

extern "C" int dmh_sampler_seek(int32_t* cursor, int k, const DmhStep* table, const int64_t* times, int S, DmhStep* cur,
                                int64_t* tcond, int B, void* stream) {
  DMH_REQUIRE(cursor && table && times && cur && tcond && S > 0 && B > 0 && k < S, "dmh_sampler_seek: bad arguments");
  hipLaunchKernelGGL(sampler_seek_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cursor, k, table, times, S, cur,
                     tcond, B);
  DMH_CHECK_LAUNCH("dmh_sampler_seek");
  return DMH_OK;
}

extern "C" int dmh_affine(const float* x, float* y, float scale, float shift, int64_t n, void* stream) {
  DMH_REQUIRE(x && y && n > 0, "dmh_affine: bad arguments");
  hipLaunchKernelGGL(affine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, scale, shift, n);
  DMH_CHECK_LAUNCH("dmh_affine");
  return DMH_OK;
}

extern "C" int dmh_affine_tail(float* x, int B, int C, int HW, int c0, float scale, float shift, void* stream) {
  DMH_REQUIRE(x && B > 0 && C > 0 && HW > 0 && c0 >= 0 && c0 < C, "dmh_affine_tail: bad arguments");
  const int64_t total = (int64_t)B * (C - c0) * HW;
  hipLaunchKernelGGL(affine_tail_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, C, HW, c0, scale,
                     shift, total);
  DMH_CHECK_LAUNCH("dmh_affine_tail");
  return DMH_OK;
}

extern "C" int dmh_q_sample(const float* x_start, const float* noise, const float* ca, const float* cb, float* out,
                            int B, int64_t per_sample, void* stream) {
  DMH_REQUIRE(x_start && noise && ca && cb && out && B > 0 && per_sample > 0, "dmh_q_sample: bad arguments");
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x_start, noise, ca, cb,
                     out, per_sample, total);
  DMH_CHECK_LAUNCH("dmh_q_sample");
  return DMH_OK;
}

extern "C" int dmh_diff_mean(const float* a, const float* b, const float* m, int squared, double* ws, float* out, int B,
                             int C, int HW, void* stream) {
  DMH_REQUIRE(a && b && ws && out && B > 0 && C > 0 && HW > 0, "dmh_diff_mean: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(diff_partial_kernel, dim3(64, B), dim3(256), 0, st, a, b, m, squared, ws, C, HW);
  DMH_CHECK_LAUNCH("dmh_diff_mean(partial)");
  hipLaunchKernelGGL(diff_final_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, ws, out, B, 1.0 / ((double)C * HW));
  DMH_CHECK_LAUNCH("dmh_diff_mean(final)");
  return DMH_OK;
}

extern "C" int dmh_loss_combine(const float* l, const float* photo, const float* w, float* out, int B, void* stream) {
  DMH_REQUIRE(l && photo && w && out && B > 0, "dmh_loss_combine: bad arguments");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, l, photo, w, out, B);
  DMH_CHECK_LAUNCH("dmh_loss_combine");
  return DMH_OK;
}

extern "C" int dmh_to_uint8(const float* img, uint8_t* out, int64_t n, void* stream) {
  DMH_REQUIRE(img && out && n > 0, "dmh_to_uint8: bad arguments");
  hipLaunchKernelGGL(to_uint8_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, img, out, n);
  DMH_CHECK_LAUNCH("dmh_to_uint8");
  return DMH_OK;
}

extern "C" int dmh_rows_lincomb(const float* x, const float* y, const float* ca, const float* cb, const float* dv, float* out,
                                int B, int64_t per_sample, int clamp, void* stream) {
  DMH_REQUIRE(x && ca && out && B > 0 && per_sample > 0 && (!y || cb), "dmh_rows_lincomb: bad arguments");
  const int64_t total = (int64_t)B * per_sample;
  hipLaunchKernelGGL(rows_lincomb_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, ca, cb, dv, out,
                     per_sample, total, clamp);
  DMH_CHECK_LAUNCH("dmh_rows_lincomb");
  return DMH_OK;
}
