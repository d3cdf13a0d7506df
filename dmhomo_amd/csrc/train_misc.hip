// Training: the small kernels around the big ones.
//   act        SiLU / GELU(erf) forward and backward on the embedding MLPs (CFG:353,362,220)
//   embedding  backward of classes_emb lookup + null-class select (CFG:419-425)
//   loss       gradient of p_losses wrt the UNet output (CFG:796-806): L1 / L2 term + the alpha_bar-weighted, masked
//              photometric term, including the transpose of flow_warp's bilinear gather (grid_sample backward wrt input)
//   optimiser  global gradient norm (clip_grad_norm_, DDP:1852), Adam (torch.optim.Adam semantics), EMA lerp
#include "common.h"

// mode 1 SiLU, 2 GELU (exact, erf)
__device__ __forceinline__ float act_fwd(float x, int mode) {
  if (mode == 1) return x / (1.0f + expf(-x));
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float act_grad(float x, int mode) {
  if (mode == 1) {
    const float sg = 1.0f / (1.0f + expf(-x));
    return sg * (1.0f + x * (1.0f - sg));
  }
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
__global__ void act_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, int64_t n,
                           int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = dy ? dy[i] * act_grad(x[i], mode) : act_fwd(x[i], mode);
}

// dtable[cls][j] += sum over kept rows of that class (row order), dnull[j] = sum over dropped rows; one thread per column
__global__ void class_embed_bwd_kernel(const float* __restrict__ d, const int64_t* __restrict__ classes,
                                       const unsigned char* __restrict__ keep, float* __restrict__ dtable,
                                       float* __restrict__ dnull, int B, int D, int num_classes) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D) return;
  for (int c = 0; c < num_classes; ++c) dtable[(size_t)c * D + j] = 0.f;
  float dn = 0.f;
  for (int b = 0; b < B; ++b) {
    const float v = d[(size_t)b * D + j];
    if (keep[b])
      dtable[(size_t)classes[b] * D + j] += v;
    else
      dn += v;
  }
  dnull[j] = dn;
}

// ---- loss backward.  out, target: [B][6][H][W]; warped = flow_warp(out[:,3:6], flow): [B][3][H][W]; mask [B][1][H][W]
// dout = d loss / d out, loss = mean_b mean_chw l(out - target) + mean_b abar[b] * mean_{3hw} mask * l(warped - im1)
__global__ __launch_bounds__(256) void loss_bwd_direct_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                              const float* __restrict__ warped,
                                                              const float* __restrict__ mask, const float* __restrict__ abar,
                                                              float* __restrict__ dout, float* __restrict__ gD, int B, int HW,
                                                              int squared) {
  // every element of dout gets its direct term; gD = d loss / d (warped - im1) is kept for the scatter pass
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * 6 * HW;
  if (i >= total) return;
  const int b = (int)(i / ((int64_t)6 * HW));
  const int c = (int)((i / HW) % 6);
  const int p = (int)(i % HW);
  const float diff = out[i] - target[i];
  const float n1 = 1.0f / ((float)B * 6.0f * (float)HW);
  float g = squared ? 2.0f * diff * n1 : (diff > 0.f ? n1 : (diff < 0.f ? -n1 : 0.f));
  if (c < 3) {  // im1 = out[:, c]: d/d im1 of mask*l(warped - im1)
    const size_t j = ((size_t)b * 3 + c) * HW + p;
    const float D = warped[j] - out[i];
    const float n2 = abar[b] * mask[(size_t)b * HW + p] / ((float)B * 3.0f * (float)HW);
    const float gd = squared ? 2.0f * D * n2 : (D > 0.f ? n2 : (D < 0.f ? -n2 : 0.f));
    gD[j] = gd;
    g -= gd;
  }
  dout[i] = g;
}

// transpose of flow_warp_kernel's gather: dout[b][3+c][corner] += weight * gD[b][c][p]   (float atomics: the one place
// in the library where the summation order is not fixed; corners collide only where the flow field folds)
__global__ __launch_bounds__(256) void flow_warp_bwd_kernel(const float* __restrict__ gD, const float* __restrict__ flow,
                                                            float* __restrict__ dout, int H, int W) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int yi = p / W, xi = p % W;
  const size_t hw = (size_t)H * W;
  const float vx = (float)xi + flow[((size_t)b * 2 + 0) * hw + p];
  const float vy = (float)yi + flow[((size_t)b * 2 + 1) * hw + p];
  const float gx = 2.0f * vx / (float)(W - 1) - 1.0f;
  const float gy = 2.0f * vy / (float)(H - 1) - 1.0f;
  float ix = (gx + 1.f) * ((float)(W - 1) / 2.f);
  float iy = (gy + 1.f) * ((float)(H - 1) / 2.f);
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float w = ix - fx0, e = (fx0 + 1.f) - ix, n = iy - fy0, s = (fy0 + 1.f) - iy;
  const bool x1ok = x0 + 1 <= W - 1, y1ok = y0 + 1 <= H - 1;
  for (int c = 0; c < 3; ++c) {
    const float g = gD[((size_t)b * 3 + c) * hw + p];
    if (g == 0.f) continue;
    float* dc = dout + ((size_t)b * 6 + 3 + c) * hw;
    atomicAdd(dc + (size_t)y0 * W + x0, g * (s * e));
    if (x1ok) atomicAdd(dc + (size_t)y0 * W + x0 + 1, g * (s * w));
    if (y1ok) atomicAdd(dc + (size_t)(y0 + 1) * W + x0, g * (n * e));
    if (x1ok && y1ok) atomicAdd(dc + (size_t)(y0 + 1) * W + x0 + 1, g * (n * w));
  }
}

// ---- optimiser
// partial sums of squares: part[block] (f64), fixed order inside a block
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)g[i] * g[i];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// total = sum of part[0..n): thread t adds part[t], part[t+256], ... then a fixed tree -> the same value on every run;
// norm_out[0] = sqrt(total), norm_out[1] = clip coefficient min(1, max_norm/(norm+1e-6))
__global__ __launch_bounds__(256) void gradnorm_finalize_kernel(const double* __restrict__ part, int n, float max_norm,
                                                                float* __restrict__ norm_out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const float nrm = (float)sqrt(red[0]);
  norm_out[0] = nrm;
  const float c = max_norm / (nrm + 1e-6f);
  norm_out[1] = c < 1.f ? c : 1.f;
}
// torch.optim.Adam (no amsgrad, no weight decay), gradient pre-scaled by gscale[1] (the clip coefficient, device scalar)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ gscale, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gr = g[i] * (gscale ? gscale[1] : 1.f);
  const float mi = b1 * m[i] + (1.f - b1) * gr;
  const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}
// ema = ema + (1 - decay) * (p - ema)
__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float decay) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ema[i] += (1.f - decay) * (p[i] - ema[i]);
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int dmh_act(const float* x, const float* dy, float* out, int64_t n, int mode, void* stream) {
  DMH_REQUIRE(x && out && n > 0 && (mode == 1 || mode == 2), "dmh_act: bad arguments");
  hipLaunchKernelGGL(act_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n, mode);
  DMH_CHECK_LAUNCH("dmh_act");
  return DMH_OK;
}
extern "C" int dmh_class_embed_backward(const float* d, const int64_t* classes, const unsigned char* keep, float* dtable,
                                        float* dnull, int B, int D, int num_classes, void* stream) {
  DMH_REQUIRE(d && classes && keep && dtable && dnull && B > 0 && D > 0 && num_classes > 0, "dmh_class_embed_backward: bad arguments");
  hipLaunchKernelGGL(class_embed_bwd_kernel, dim3(cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, d, classes, keep, dtable,
                     dnull, B, D, num_classes);
  DMH_CHECK_LAUNCH("dmh_class_embed_backward");
  return DMH_OK;
}
// dout [B][6][H][W]; gD [B][3][H][W] work
extern "C" int dmh_loss_backward(const float* out, const float* target, const float* warped, const float* mask,
                                 const float* flow, const float* abar, float* dout, float* gD, int B, int H, int W,
                                 int squared, void* stream) {
  DMH_REQUIRE(out && target && warped && mask && flow && abar && dout && gD && B > 0 && H > 1 && W > 1,
              "dmh_loss_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * 6 * H * W;
  hipLaunchKernelGGL(loss_bwd_direct_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, out, target, warped, mask,
                     abar, dout, gD, B, H * W, squared);
  DMH_CHECK_LAUNCH("dmh_loss_backward(direct)");
  hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, st, gD, flow, dout, H, W);
  DMH_CHECK_LAUNCH("dmh_loss_backward(warp)");
  return DMH_OK;
}
#define DMH_SUMSQ_BLOCKS 256
extern "C" int dmh_sumsq_blocks(void) { return DMH_SUMSQ_BLOCKS; }
// part: f64 [dmh_sumsq_blocks()] for this tensor
extern "C" int dmh_sumsq(const float* g, int64_t n, double* part, void* stream) {
  DMH_REQUIRE(g && part && n > 0, "dmh_sumsq: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(DMH_SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, n, part);
  DMH_CHECK_LAUNCH("dmh_sumsq");
  return DMH_OK;
}
extern "C" int dmh_gradnorm_finalize(const double* part, int n, float max_norm, float* norm_out, void* stream) {
  DMH_REQUIRE(part && norm_out && n > 0, "dmh_gradnorm_finalize: bad arguments");
  hipLaunchKernelGGL(gradnorm_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, n, max_norm, norm_out);
  DMH_CHECK_LAUNCH("dmh_gradnorm_finalize");
  return DMH_OK;
}
extern "C" int dmh_adam(float* p, const float* g, float* m, float* v, const float* gscale, int64_t n, float lr, float b1,
                        float b2, float eps, int step, void* stream) {
  DMH_REQUIRE(p && g && m && v && n > 0 && step > 0, "dmh_adam: bad arguments");
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, gscale, n, lr,
                     b1, b2, eps, bc1, bc2);
  DMH_CHECK_LAUNCH("dmh_adam");
  return DMH_OK;
}
// ---- multi-tensor forms: the ~280 parameter tensors of the UNet in a handful of launches.  A launch carries up to
// MT_MAX tensors in its kernel arguments (pointers, lengths, first block of each); a block finds its tensor by binary
// search and handles MT_CHUNK consecutive elements of it.
#define MT_MAX 48
#define MT_CHUNK 4096
struct MtTable {
  float* p[MT_MAX];
  const float* g[MT_MAX];
  float* m[MT_MAX];
  float* v[MT_MAX];
  long long n[MT_MAX];
  int blk0[MT_MAX + 1];
  int count;
};
__device__ __forceinline__ int mt_find(const MtTable& t, int blk) {
  int lo = 0, hi = t.count;  // blk0[lo] <= blk < blk0[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.blk0[mid] <= blk) lo = mid; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void sumsq_multi_kernel(MtTable t, double* __restrict__ part) {
  __shared__ double red[4];
  const int ti = mt_find(t, blockIdx.x);
  const long long base = (long long)(blockIdx.x - t.blk0[ti]) * MT_CHUNK;
  const float* g = t.g[ti];
  const long long end = min(base + MT_CHUNK, t.n[ti]);
  double s = 0.0;
  for (long long i = base + threadIdx.x; i < end; i += 256) s += (double)g[i] * g[i];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void adam_multi_kernel(MtTable t, const float* __restrict__ gscale, float lr, float b1,
                                                         float b2, float eps, float bc1, float bc2) {
  const int ti = mt_find(t, blockIdx.x);
  const long long base = (long long)(blockIdx.x - t.blk0[ti]) * MT_CHUNK;
  const long long end = min(base + MT_CHUNK, t.n[ti]);
  float* p = t.p[ti];
  const float* g = t.g[ti];
  float* m = t.m[ti];
  float* v = t.v[ti];
  const float gs = gscale ? gscale[1] : 1.f;
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    const float gr = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}
// blocks dmh_sumsq_multi writes partials for (the size of ``part``)
extern "C" int64_t dmh_multi_blocks(const int64_t* n, int count) {
  if (!n || count <= 0) return count == 0 ? 0 : -1;
  int64_t b = 0;
  for (int i = 0; i < count; ++i) b += cdiv64(n[i], MT_CHUNK);
  return b;
}
static int mt_fill(MtTable& t, int i0, int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                   const int64_t* n) {
  int blocks = 0, k = 0;
  for (; k < MT_MAX && i0 + k < count; ++k) {
    t.p[k] = p ? p[i0 + k] : nullptr;
    t.g[k] = g[i0 + k];
    t.m[k] = m ? m[i0 + k] : nullptr;
    t.v[k] = v ? v[i0 + k] : nullptr;
    t.n[k] = n[i0 + k];
    t.blk0[k] = blocks;
    blocks += (int)cdiv64(n[i0 + k], MT_CHUNK);
  }
  t.blk0[k] = blocks;
  t.count = k;
  return blocks;
}
// g, n: HOST arrays of ``count`` device pointers / lengths; part: f64 [dmh_multi_blocks(n, count)]
extern "C" int dmh_sumsq_multi(const float* const* g, const int64_t* n, int count, double* part, void* stream) {
  DMH_REQUIRE(g && n && part && count > 0, "dmh_sumsq_multi: bad arguments");
  int64_t done = 0;
  for (int i0 = 0; i0 < count; i0 += MT_MAX) {
    MtTable t;
    const int blocks = mt_fill(t, i0, count, nullptr, g, nullptr, nullptr, n);
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, part + done);
    done += blocks;
  }
  DMH_CHECK_LAUNCH("dmh_sumsq_multi");
  return DMH_OK;
}
extern "C" int dmh_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                              int count, const float* gscale, float lr, float b1, float b2, float eps, int step,
                              void* stream) {
  DMH_REQUIRE(p && g && m && v && n && count > 0 && step > 0, "dmh_adam_multi: bad arguments");
  const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
  for (int i0 = 0; i0 < count; i0 += MT_MAX) {
    MtTable t;
    const int blocks = mt_fill(t, i0, count, p, g, m, v, n);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, gscale, lr, b1, b2, eps, bc1,
                       bc2);
  }
  DMH_CHECK_LAUNCH("dmh_adam_multi");
  return DMH_OK;
}
extern "C" int dmh_ema(float* ema, const float* p, int64_t n, float decay, void* stream) {
  DMH_REQUIRE(ema && p && n > 0, "dmh_ema: bad arguments");
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, ema, p, n, decay);
  DMH_CHECK_LAUNCH("dmh_ema");
  return DMH_OK;
}
