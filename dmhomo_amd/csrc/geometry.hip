// K7-K9 — condition builder & geometry: homography -> flow -> HSV image, bilinear flow warp,
// flow -> homography (DLT normal equations).  Mirrors the reference's op order exactly where
// integers come out (grid-sample corner indices): no FMA contraction in this file.
#include "common.h"

#pragma clang fp contract(off)

// G3: flow_to_image DDP:1479-1485 + matplotlib.colors.hsv_to_rgb for one pixel
__device__ __forceinline__ void flow_pixel_to_rgb(float u, float v, float max_flow, float& r, float& g, float& bl) {
  const float n = 8.f;
  const float mag = sqrtf(u * u + v * v);
  const float ang = atan2f(v, u);
  float hh = fmodf(ang / 6.283185307179586f + 1.f, 1.f);  // np.mod(angle / (2 pi) + 1, 1), operand >= 0.5
  float ss = fminf(fmaxf(mag * n / max_flow, 0.f), 1.f);
  float vv = fminf(fmaxf(n - ss, 0.f), 1.f);
  // matplotlib.colors.hsv_to_rgb: i = (h*6).astype(int); f = h*6 - i is float64 there (f32 - int64),
  // so q and t are formed in f64 and rounded to fp32 on store; p stays fp32.
  const float h6 = hh * 6.0f;
  const int i = (int)h6;
  const double f = (double)h6 - (double)i;
  const float pp = vv * (1.0f - ss);
  const float qq = (float)((double)vv * (1.0 - (double)ss * f));
  const float tt = (float)((double)vv * (1.0 - (double)ss * (1.0 - f)));
  switch (i % 6) {
    case 0: r = vv; g = tt; bl = pp; break;
    case 1: r = qq; g = vv; bl = pp; break;
    case 2: r = pp; g = vv; bl = tt; break;
    case 3: r = pp; g = qq; bl = vv; break;
    case 4: r = tt; g = pp; bl = vv; break;
    default: r = vv; g = pp; bl = qq; break;
  }
  if (ss == 0.f) r = g = bl = vv;
}

// ---------------------------------------------------------------------------------------------
// G2 + G3.  get_flow_np DDP:954-967 in float64 on an integer grid, cast to fp32; then
// flow_to_image DDP:1479-1485 + matplotlib hsv_to_rgb in fp32.
__global__ __launch_bounds__(256) void homography_flow_kernel(const double* __restrict__ Hm, float* __restrict__ flow,
                                                              float* __restrict__ rgb, int H, int W, float max_flow) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int yi = p / W, xi = p % W;
  const double* h = Hm + b * 9;
  const double x = (double)xi, y = (double)yi;
  const double qx = h[0] * x + h[1] * y + h[2];
  const double qy = h[3] * x + h[4] * y + h[5];
  const double qw = (h[6] * x + h[7] * y + h[8]) + 1e-6;  // DDP:958-959
  const float u = (float)(qx / qw - x);
  const float v = (float)(qy / qw - y);
  const size_t hw = (size_t)H * W;
  if (flow) {
    flow[((size_t)b * 2 + 0) * hw + p] = u;
    flow[((size_t)b * 2 + 1) * hw + p] = v;
  }
  if (rgb) {
    float r, g, bl;
    flow_pixel_to_rgb(u, v, max_flow, r, g, bl);
    rgb[((size_t)b * 3 + 0) * hw + p] = r;
    rgb[((size_t)b * 3 + 1) * hw + p] = g;
    rgb[((size_t)b * 3 + 2) * hw + p] = bl;
  }
}

// ---------------------------------------------------------------------------------------------
// G4.  flow_warp = grid_sample(bilinear, border, align_corners=True) of torch's CPU kernel:
//   g  = 2.0*v/(W-1) - 1.0;  ix = (g+1)*((W-1)/2);  ix = min(W-1, max(ix, 0));  x0 = floor(ix)
//   w = ix-x0, e = (x0+1)-ix, n = iy-y0, s = (y0+1)-iy
//   out = fma(se, n*w, fma(sw, n*e, fma(ne, s*w, nw*(s*e))))
// pad: 0 'border' (the reference's default: coordinates clipped to the image), 1 'zeros' (taps outside the image read 0),
// 2 'reflection' (reflected about the border pixels' centres, then clipped) — grid_sample's padding_mode with
// align_corners=True (ATen/native/GridSampler.h: clip_coordinates, reflect_coordinates).  mode: 0 'bilinear', 1 'nearest'
// (nearbyint, ties to even), 2 'bicubic' (cubic convolution, A = -0.75, 4 x 4 taps around floor of the UNPADDED coordinate;
// the padding rule is applied to every tap index: get_value_bounded).
__device__ __forceinline__ float gs_reflect(float in, int size) {  // reflect_coordinates(in, 0, 2 * (size - 1))
  if (size <= 1) return 0.f;
  const float span = (float)(size - 1);
  in = fabsf(in);
  const float extra = fmodf(in, span);
  const int flips = (int)floorf(in / span);
  return (flips & 1) == 0 ? extra : span - extra;
}
__device__ __forceinline__ float gs_pad(float i, int size, int pad) {   // compute_coordinates
  if (pad == 2) i = gs_reflect(i, size);
  if (pad != 1) i = fminf((float)(size - 1), fmaxf(i, 0.f));
  return i;
}
__device__ __forceinline__ float gs_coord(float g, int size, int pad) {
  return gs_pad((g + 1.f) * ((float)(size - 1) / 2.f), size, pad);
}
// get_cubic_upsample_coefficients (ATen/native/UpSample.h), A = -0.75
__device__ __forceinline__ void gs_cubic(float t, float (&c)[4]) {
  const float A = -0.75f;
  auto c1 = [A](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
  auto c2 = [A](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
  c[0] = c1(t + 1.f);
  c[1] = c2(t);
  c[2] = c2(1.f - t);
  c[3] = c1(2.f - t);
}
__global__ __launch_bounds__(256) void flow_warp_general_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                                float* __restrict__ out, int C, int H, int W, int pad, int mode) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int yi = p / W, xi = p % W;
  const size_t hw = (size_t)H * W;
  const float vx = (float)xi + flow[((size_t)b * 2 + 0) * hw + p];
  const float vy = (float)yi + flow[((size_t)b * 2 + 1) * hw + p];
  const float gx = 2.0f * vx / (float)(W - 1) - 1.0f;
  const float gy = 2.0f * vy / (float)(H - 1) - 1.0f;
  if (mode == 2) {
    const float ux = (gx + 1.f) * ((float)(W - 1) / 2.f), uy = (gy + 1.f) * ((float)(H - 1) / 2.f);   // unnormalize only
    const float fx0 = floorf(ux), fy0 = floorf(uy);
    float cx[4], cy[4];
    gs_cubic(ux - fx0, cx);
    gs_cubic(uy - fy0, cy);
    for (int c = 0; c < C; ++c) {
      const float* xc = x + ((size_t)b * C + c) * hw;
      float acc = 0.f;
      for (int i = 0; i < 4; ++i) {
        const float py = gs_pad(fy0 - 1.f + (float)i, H, pad);
        float row = 0.f;
        for (int j = 0; j < 4; ++j) {
          const float px = gs_pad(fx0 - 1.f + (float)j, W, pad);
          const bool ok = px >= 0.f && px <= (float)(W - 1) && py >= 0.f && py <= (float)(H - 1);
          row += (ok ? xc[(size_t)(int)py * W + (int)px] : 0.f) * cx[j];
        }
        acc += row * cy[i];
      }
      out[((size_t)b * C + c) * hw + p] = acc;
    }
    return;
  }
  const float ix = gs_coord(gx, W, pad), iy = gs_coord(gy, H, pad);
  auto tap = [&](const float* xc, float fx, float fy) -> float {   // within_bounds_2d ? value : 0
    // (a NaN or far-away coordinate compares false / falls outside: 0, as safe_get / the masked gather give)
    if (!(fx >= 0.f && fx <= (float)(W - 1) && fy >= 0.f && fy <= (float)(H - 1))) return 0.f;
    return xc[(size_t)(int)fy * W + (int)fx];
  };
  if (mode == 1) {
    const float nx = nearbyintf(ix), ny = nearbyintf(iy);
    for (int c = 0; c < C; ++c) out[((size_t)b * C + c) * hw + p] = tap(x + ((size_t)b * C + c) * hw, nx, ny);
    return;
  }
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const float w = ix - fx0, e = (fx0 + 1.f) - ix, n = iy - fy0, s = (fy0 + 1.f) - iy;
  const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
  for (int c = 0; c < C; ++c) {
    const float* xc = x + ((size_t)b * C + c) * hw;
    float acc = tap(xc, fx0, fy0) * wnw;
    acc = fmaf(tap(xc, fx0 + 1.f, fy0), wne, acc);
    acc = fmaf(tap(xc, fx0, fy0 + 1.f), wsw, acc);
    acc = fmaf(tap(xc, fx0 + 1.f, fy0 + 1.f), wse, acc);
    out[((size_t)b * C + c) * hw + p] = acc;
  }
}

__global__ __launch_bounds__(256) void flow_warp_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                        float* __restrict__ out, int32_t* __restrict__ x0o,
                                                        int32_t* __restrict__ y0o, int C, int H, int W) {
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
  if (x0o) x0o[(size_t)b * hw + p] = x0;
  if (y0o) y0o[(size_t)b * hw + p] = y0;
  const float w = ix - fx0, e = (fx0 + 1.f) - ix, n = iy - fy0, s = (fy0 + 1.f) - iy;
  const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
  const bool x1ok = x0 + 1 <= W - 1, y1ok = y0 + 1 <= H - 1;
  const int x1 = x1ok ? x0 + 1 : x0, y1 = y1ok ? y0 + 1 : y0;
  for (int c = 0; c < C; ++c) {
    const float* xc = x + ((size_t)b * C + c) * hw;
    const float nw = xc[(size_t)y0 * W + x0];
    const float ne = x1ok ? xc[(size_t)y0 * W + x1] : 0.f;
    const float sw = y1ok ? xc[(size_t)y1 * W + x0] : 0.f;
    const float se = (x1ok && y1ok) ? xc[(size_t)y1 * W + x1] : 0.f;
    float acc = nw * wnw;
    acc = fmaf(ne, wne, acc);
    acc = fmaf(sw, wsw, acc);
    acc = fmaf(se, wse, acc);
    out[((size_t)b * C + c) * hw + p] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// G5.  DLT_solve DDP:1612-1643 for one homography per sample: least squares over all pixels,
//   [x y 1 0 0 0 -x'x -x'y] h = x',  [0 0 0 x y 1 -y'x -y'y] h = y',  (x',y') = (x,y) + flow.
// pinv(A) b == (A^T A)^-1 A^T b for full column rank; A^T A (36 unique) and A^T b (8) are
// accumulated in f64 per block (fixed order), then a second kernel sums the blocks and solves
// the column-equilibrated 8x8 system by Cholesky.
__device__ __forceinline__ void dlt_accum(double* acc, const double* r, double rhs) {
  int k = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = i; j < 8; ++j) acc[k++] += r[i] * r[j];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[36 + i] += r[i] * rhs;
}

__global__ __launch_bounds__(256) void dlt_accumulate_kernel(const float* __restrict__ flow, double* __restrict__ ws,
                                                             int H, int W) {
  __shared__ double red[4][44];
  const int b = blockIdx.y, blk = blockIdx.x;
  const size_t hw = (size_t)H * W;
  double acc[44];
#pragma unroll
  for (int i = 0; i < 44; ++i) acc[i] = 0.0;
  for (int p = blk * 256 + threadIdx.x; p < (int)hw; p += DMH_DLT_BLOCKS * 256) {
    const double x = (double)(p % W), y = (double)(p / W);
    const double xd = x + (double)flow[((size_t)b * 2 + 0) * hw + p];
    const double yd = y + (double)flow[((size_t)b * 2 + 1) * hw + p];
    const double ru[8] = {x, y, 1.0, 0.0, 0.0, 0.0, -xd * x, -xd * y};
    const double rv[8] = {0.0, 0.0, 0.0, x, y, 1.0, -yd * x, -yd * y};
    dlt_accum(acc, ru, xd);
    dlt_accum(acc, rv, yd);
  }
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    double v = acc[i];
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    acc[i] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 44; ++i) red[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 44)
    ws[((size_t)b * DMH_DLT_BLOCKS + blk) * 44 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void dlt_solve_kernel(const double* __restrict__ ws, double* __restrict__ Hout, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s[44];
  for (int i = 0; i < 44; ++i) s[i] = 0.0;
  for (int blk = 0; blk < DMH_DLT_BLOCKS; ++blk)
    for (int i = 0; i < 44; ++i) s[i] += ws[((size_t)b * DMH_DLT_BLOCKS + blk) * 44 + i];
  double G[8][8], r[8], d[8];
  int k = 0;
  for (int i = 0; i < 8; ++i)
    for (int j = i; j < 8; ++j) {
      G[i][j] = s[k];
      G[j][i] = s[k];
      ++k;
    }
  for (int i = 0; i < 8; ++i) d[i] = 1.0 / sqrt(G[i][i]);
  for (int i = 0; i < 8; ++i) {
    r[i] = s[36 + i] * d[i];
    for (int j = 0; j < 8; ++j) G[i][j] *= d[i] * d[j];
  }
  // Cholesky G = L L^T (in the lower triangle)
  for (int j = 0; j < 8; ++j) {
    double v = G[j][j];
    for (int q = 0; q < j; ++q) v -= G[j][q] * G[j][q];
    const double ljj = sqrt(v);
    G[j][j] = ljj;
    for (int i = j + 1; i < 8; ++i) {
      double t = G[i][j];
      for (int q = 0; q < j; ++q) t -= G[i][q] * G[j][q];
      G[i][j] = t / ljj;
    }
  }
  double yv[8], hv[8];
  for (int i = 0; i < 8; ++i) {
    double t = r[i];
    for (int q = 0; q < i; ++q) t -= G[i][q] * yv[q];
    yv[i] = t / G[i][i];
  }
  for (int i = 7; i >= 0; --i) {
    double t = yv[i];
    for (int q = i + 1; q < 8; ++q) t -= G[q][i] * hv[q];
    hv[i] = t / G[i][i];
  }
  for (int i = 0; i < 8; ++i) Hout[(size_t)b * 9 + i] = hv[i] * d[i];
  Hout[(size_t)b * 9 + 8] = 1.0;
}

extern "C" int dmh_homography_flow(const double* Hm, float* flow, float* rgb, int B, int H, int W, float max_flow,
                                   void* stream) {
  DMH_REQUIRE(Hm && (flow || rgb) && B > 0 && H > 0 && W > 0, "dmh_homography_flow: bad arguments");
  hipLaunchKernelGGL(homography_flow_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, Hm, flow,
                     rgb, H, W, max_flow);
  DMH_CHECK_LAUNCH("dmh_homography_flow");
  return DMH_OK;
}

__global__ __launch_bounds__(256) void flow_image_kernel(const float* __restrict__ flow, float* __restrict__ rgb,
                                                         int HW, float max_flow) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float r, g, bl;
  flow_pixel_to_rgb(flow[((size_t)b * 2 + 0) * HW + p], flow[((size_t)b * 2 + 1) * HW + p], max_flow, r, g, bl);
  rgb[((size_t)b * 3 + 0) * HW + p] = r;
  rgb[((size_t)b * 3 + 1) * HW + p] = g;
  rgb[((size_t)b * 3 + 2) * HW + p] = bl;
}

extern "C" int dmh_flow_to_image(const float* flow, float* rgb, int B, int HW, float max_flow, void* stream) {
  DMH_REQUIRE(flow && rgb && B > 0 && HW > 0, "dmh_flow_to_image: bad arguments");
  hipLaunchKernelGGL(flow_image_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, flow, rgb, HW,
                     max_flow);
  DMH_CHECK_LAUNCH("dmh_flow_to_image");
  return DMH_OK;
}

extern "C" int dmh_flow_warp(const float* x, const float* flow, float* out, int32_t* x0, int32_t* y0, int B, int C,
                             int H, int W, int pad, int mode, void* stream) {
  DMH_REQUIRE(x && flow && out && B > 0 && C > 0 && H > 1 && W > 1, "dmh_flow_warp: bad arguments");
  DMH_REQUIRE(pad >= 0 && pad <= 2 && mode >= 0 && mode <= 2,
              "dmh_flow_warp: pad=%d (0 border, 1 zeros, 2 reflection), mode=%d (0 bilinear, 1 nearest, 2 bicubic)", pad, mode);
  if (pad != 0 || mode != 0) {
    DMH_REQUIRE(!x0 && !y0, "dmh_flow_warp: the corner indices are an output of the default (border, bilinear) form only");
    hipLaunchKernelGGL(flow_warp_general_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, x, flow, out, C,
                       H, W, pad, mode);
    DMH_CHECK_LAUNCH("dmh_flow_warp");
    return DMH_OK;
  }
  hipLaunchKernelGGL(flow_warp_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, x, flow, out, x0,
                     y0, C, H, W);
  DMH_CHECK_LAUNCH("dmh_flow_warp");
  return DMH_OK;
}

extern "C" int dmh_dlt_homography(const float* flow, double* ws, double* Hout, int B, int H, int W, void* stream) {
  DMH_REQUIRE(flow && ws && Hout && B > 0 && H > 0 && W > 0, "dmh_dlt_homography: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dlt_accumulate_kernel, dim3(DMH_DLT_BLOCKS, B), dim3(256), 0, st, flow, ws, H, W);
  DMH_CHECK_LAUNCH("dmh_dlt_homography(accumulate)");
  hipLaunchKernelGGL(dlt_solve_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, ws, Hout, B);
  DMH_CHECK_LAUNCH("dmh_dlt_homography(solve)");
  return DMH_OK;
}

// ---------------------------------------------------------------------------------------------
// The helper functions around G2 / G4 / G5 that scripts import by name: get_grid DDP:1558-1574,
// norm_grid DDP:1292-1299, the multi-band / arbitrary-index get_flow_np DDP:927-969, DLT_solve DDP:1577-1644 on explicit
// point sets.
__global__ __launch_bounds__(256) void pixel_grid_kernel(float* __restrict__ out, int H, int W, float start) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const size_t hw = (size_t)H * W;
  out[((size_t)b * 2 + 0) * hw + p] = (float)(p % W) + start;
  out[((size_t)b * 2 + 1) * hw + p] = (float)(p / W) + start;
}

// v (B,2,H,W) -> (B,H,W,2): 2.0 * v / (W-1) - 1.0 (x), 2.0 * v / (H-1) - 1.0 (y), fp32 in that op order
__global__ __launch_bounds__(256) void norm_grid_kernel(const float* __restrict__ v, float* __restrict__ out, int H, int W) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const size_t hw = (size_t)H * W;
  const float gx = 2.0f * v[((size_t)b * 2 + 0) * hw + p] / (float)(W - 1) - 1.0f;
  const float gy = 2.0f * v[((size_t)b * 2 + 1) * hw + p] / (float)(H - 1) - 1.0f;
  out[((size_t)b * hw + p) * 2 + 0] = gx;
  out[((size_t)b * hw + p) * 2 + 1] = gy;
}

// get_flow_np: Hm (B, divide, 3, 3) f64, idx (Bi, 3, H, W) f64 (Bi == B or 1), flow (B, 2, H, W) f64.  Row y uses the
// homography of band min(y / (H / divide), divide - 1) (DDP:940-951); w' += 1e-6 unconditionally (DDP:958-959).
__global__ __launch_bounds__(256) void homography_flow_points_kernel(const double* __restrict__ Hm, const double* __restrict__ idx,
                                                                     double* __restrict__ flow, int divide, int Bi, int H,
                                                                     int W) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const size_t hw = (size_t)H * W;
  const int band_h = H / divide;
  int band = band_h > 0 ? (p / W) / band_h : 0;
  band = band < divide - 1 ? band : divide - 1;
  const double* h = Hm + ((size_t)b * divide + band) * 9;
  const double* ip = idx + (size_t)(Bi == 1 ? 0 : b) * 3 * hw + p;
  const double x = ip[0], y = ip[hw], o = ip[2 * hw];
  const double qx = h[0] * x + h[1] * y + h[2] * o;
  const double qy = h[3] * x + h[4] * y + h[5] * o;
  const double qw = (h[6] * x + h[7] * y + h[8] * o) + 1e-6;
  flow[((size_t)b * 2 + 0) * hw + p] = qx / qw - x;
  flow[((size_t)b * 2 + 1) * hw + p] = qy / qw - y;
}

// DLT on explicit correspondences: src, off (N, P, 2) f64; system n: least squares over its P points (dst = src + off)
__global__ __launch_bounds__(256) void dlt_accumulate_points_kernel(const double* __restrict__ src, const double* __restrict__ off,
                                                                    double* __restrict__ ws, int P) {
  __shared__ double red[4][44];
  const int b = blockIdx.y, blk = blockIdx.x;
  double acc[44];
#pragma unroll
  for (int i = 0; i < 44; ++i) acc[i] = 0.0;
  for (int p = blk * 256 + threadIdx.x; p < P; p += DMH_DLT_BLOCKS * 256) {
    const size_t q = ((size_t)b * P + p) * 2;
    const double x = src[q], y = src[q + 1];
    const double xd = x + off[q], yd = y + off[q + 1];
    const double ru[8] = {x, y, 1.0, 0.0, 0.0, 0.0, -xd * x, -xd * y};
    const double rv[8] = {0.0, 0.0, 0.0, x, y, 1.0, -yd * x, -yd * y};
    dlt_accum(acc, ru, xd);
    dlt_accum(acc, rv, yd);
  }
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    double v = acc[i];
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    acc[i] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 44; ++i) red[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 44)
    ws[((size_t)b * DMH_DLT_BLOCKS + blk) * 44 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// P == 4: the 8x8 system is square — solved as it stands (Gaussian elimination with partial pivoting, f64) instead of
// through its normal equations, which would square the condition number of an exactly determined system
__global__ void dlt_solve4_kernel(const double* __restrict__ src, const double* __restrict__ off, double* __restrict__ Hout,
                                  int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double A[8][9];
  for (int p = 0; p < 4; ++p) {
    const size_t q = ((size_t)n * 4 + p) * 2;
    const double x = src[q], y = src[q + 1];
    const double xd = x + off[q], yd = y + off[q + 1];
    const double ru[9] = {x, y, 1.0, 0.0, 0.0, 0.0, -xd * x, -xd * y, xd};
    const double rv[9] = {0.0, 0.0, 0.0, x, y, 1.0, -yd * x, -yd * y, yd};
    for (int j = 0; j < 9; ++j) {
      A[2 * p][j] = ru[j];
      A[2 * p + 1][j] = rv[j];
    }
  }
  // (rank-deficient systems — repeated or collinear corners — have no pivot to divide by, while the reference's
  //  torch.linalg.pinv, DDP:1639, returns the finite minimum-norm least-squares solution: such a system is re-solved through
  //  its Tikhonov-regularised normal equations (A^T A + lambda I) h = A^T b, whose solution tends to pinv(A) b as lambda -> 0;
  //  lambda = 1e-8 * trace / 8 ~ sqrt(f64 epsilon) * |A|^2 balances the bias (lambda / sigma^2 of the retained directions)
  //  against the rounding of A^T A (epsilon * |A|^2 / lambda): ~1e-6 of the minimum-norm solution.  Well-posed systems never
  //  take this path.)
  double amax = 0.0;
  for (int r = 0; r < 8; ++r)
    for (int j = 0; j < 8; ++j) amax = fmax(amax, fabs(A[r][j]));
  double M[8][9];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 9; ++j) {
      double t = 0.0;
      for (int r = 0; r < 8; ++r) t += A[r][i] * A[r][j];
      M[i][j] = t;
    }
  bool singular = false;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      if (!singular) break;
      double tr = 0.0;
      for (int i = 0; i < 8; ++i) tr += M[i][i];
      const double lam = (tr > 0.0 ? tr : 1.0) * 1.25e-9;
      for (int i = 0; i < 8; ++i) {
        for (int j = 0; j < 9; ++j) A[i][j] = M[i][j];
        A[i][i] += lam;
      }
      amax = 0.0;   // (the regularised system is positive definite: solve it whatever its pivots are)
    }
    for (int c = 0; c < 8; ++c) {
      int piv = c;
      for (int r = c + 1; r < 8; ++r)
        if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
      for (int j = 0; j < 9; ++j) {
        const double t = A[c][j];
        A[c][j] = A[piv][j];
        A[piv][j] = t;
      }
      const double d = A[c][c];
      if (!(fabs(d) > 1e-13 * amax)) {
        singular = true;
        if (pass == 0) break;
      }
      for (int r = c + 1; r < 8; ++r) {
        const double f = A[r][c] / d;
        for (int j = c; j < 9; ++j) A[r][j] -= f * A[c][j];
      }
    }
  }
  double h[8];
  for (int i = 7; i >= 0; --i) {
    double t = A[i][8];
    for (int j = i + 1; j < 8; ++j) t -= A[i][j] * h[j];
    h[i] = t / A[i][i];
  }
  for (int i = 0; i < 8; ++i) Hout[(size_t)n * 9 + i] = h[i];
  Hout[(size_t)n * 9 + 8] = 1.0;
}

extern "C" int dmh_pixel_grid(float* out, int B, int H, int W, float start, void* stream) {
  DMH_REQUIRE(out && B > 0 && H > 0 && W > 0, "dmh_pixel_grid: bad arguments");
  hipLaunchKernelGGL(pixel_grid_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, out, H, W, start);
  DMH_CHECK_LAUNCH("dmh_pixel_grid");
  return DMH_OK;
}

extern "C" int dmh_norm_grid(const float* v, float* out, int B, int H, int W, void* stream) {
  DMH_REQUIRE(v && out && B > 0 && H > 0 && W > 0, "dmh_norm_grid: bad arguments");
  hipLaunchKernelGGL(norm_grid_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, v, out, H, W);
  DMH_CHECK_LAUNCH("dmh_norm_grid");
  return DMH_OK;
}

extern "C" int dmh_homography_flow_points(const double* Hm, const double* idx, double* flow, int B, int divide, int Bi, int H,
                                          int W, void* stream) {
  DMH_REQUIRE(Hm && idx && flow && B > 0 && divide > 0 && H >= divide && W > 0 && (Bi == 1 || Bi == B),
              "dmh_homography_flow_points: bad arguments");
  hipLaunchKernelGGL(homography_flow_points_kernel, dim3(cdiv(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, Hm, idx,
                     flow, divide, Bi, H, W);
  DMH_CHECK_LAUNCH("dmh_homography_flow_points");
  return DMH_OK;
}

extern "C" int dmh_dlt_points(const double* src, const double* off, double* ws, double* Hout, int N, int P, void* stream) {
  DMH_REQUIRE(src && off && ws && Hout && N > 0 && P >= 4, "dmh_dlt_points: bad arguments (N systems of P >= 4 points)");
  DMH_REQUIRE(N <= 65535, "dmh_dlt_points: N=%d systems (limit 65535 per call)", N);
  hipStream_t st = (hipStream_t)stream;
  if (P == 4) {
    hipLaunchKernelGGL(dlt_solve4_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, src, off, Hout, N);
    DMH_CHECK_LAUNCH("dmh_dlt_points(4-point solve)");
    return DMH_OK;
  }
  hipLaunchKernelGGL(dlt_accumulate_points_kernel, dim3(DMH_DLT_BLOCKS, N), dim3(256), 0, st, src, off, ws, P);
  DMH_CHECK_LAUNCH("dmh_dlt_points(accumulate)");
  hipLaunchKernelGGL(dlt_solve_kernel, dim3(cdiv(N, 64)), dim3(64), 0, st, ws, Hout, N);
  DMH_CHECK_LAUNCH("dmh_dlt_points(solve)");
  return DMH_OK;
}
