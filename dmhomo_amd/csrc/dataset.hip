// Condition dataset path: the per-item preprocessing of UnHomoTrainData.__getitem__ (DDP:1097-1163)
// on the device, a whole batch per launch.  The reference does this per item in DataLoader workers with OpenCV:
//
//   img  = cv2.resize(cv2.imread(png).astype(float32) / 255., (S, S))              bilinear (INTER_LINEAR)   DDP:1118-1123
//   mask = cv2.dilate(cv2.erode(cv2.resize(mask, (S, S), INTER_NEAREST), 3x3), 3x3)                         DDP:1129-1133
//
// Here the host only decodes the PNGs and hands over the uint8 pixels; the two kernels below write straight into the
// channel planes of the 12-channel batch [img1(3) img2(3) mask(1) rgb_flow(3) flow(2)] (DDP:1162), whose last five
// planes come from dmh_homography_flow.  OpenCV is not installed in the build image, so its published algorithm is
// restated (resize.cpp: pixel-centre mapping fx = (dx + 0.5) * scale - 0.5 in float, clamped taps, horizontal pass then
// vertical pass in float; resizeNN: sx = min(floor(dx * ifx), W - 1); morphology with the default border = "ignore
// pixels outside the image") — oracle/dataset.py holds the same restatement on the CPU; parity with cv2 itself is
// UNPINNED.
#include "common.h"

namespace {
__device__ __forceinline__ void lin_tap(int d, double scale, int n, int& s, float& f) {
  f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= n - 1) {
    f = 0.f;
    s = n - 1;
  }
}
}  // namespace

// src: [B][Hs][Ws][C] uint8 (interleaved, as decoded); dst plane c of image b at dst + b*bstride + c*Hd*Wd
__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                                 int B, int Hs, int Ws, int C, int Hd, int Wd, int64_t bstride,
                                                                 float div) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * Hd * Wd;
  if (i >= total) return;
  const int dx = i % Wd, dy = (i / Wd) % Hd, b = i / ((int64_t)Wd * Hd);
  int sx, sy;
  float fx, fy;
  lin_tap(dx, (double)Ws / Wd, Ws, sx, fx);
  lin_tap(dy, (double)Hs / Hd, Hs, sy, fy);
  const int sx1 = min(sx + 1, Ws - 1), sy1 = min(sy + 1, Hs - 1);
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const unsigned char* im = src + (size_t)b * Hs * Ws * C;
  for (int c = 0; c < C; ++c) {
    const float s00 = (float)im[((size_t)sy * Ws + sx) * C + c] / div, s01 = (float)im[((size_t)sy * Ws + sx1) * C + c] / div;
    const float s10 = (float)im[((size_t)sy1 * Ws + sx) * C + c] / div, s11 = (float)im[((size_t)sy1 * Ws + sx1) * C + c] / div;
    const float r0 = s00 * a0 + s01 * a1, r1 = s10 * a0 + s11 * a1;
    dst[b * bstride + ((int64_t)c * Hd + dy) * Wd + dx] = r0 * b0 + r1 * b1;
  }
}

// nearest resize + 3x3 erode + 3x3 dilate of a float mask [B][Hs][Ws] -> plane at dst + b*bstride
__global__ __launch_bounds__(256) void mask_open_nearest_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                                int Hs, int Ws, int Hd, int Wd, int64_t bstride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * Hd * Wd;
  if (i >= total) return;
  const int dx = i % Wd, dy = (i / Wd) % Hd, b = i / ((int64_t)Wd * Hd);
  const double ifx = 1.0 / ((double)Wd / Ws), ify = 1.0 / ((double)Hd / Hs);
  const float* m = src + (size_t)b * Hs * Ws;
  float best = -INFINITY;
  for (int qy = dy - 1; qy <= dy + 1; ++qy)
    for (int qx = dx - 1; qx <= dx + 1; ++qx) {
      if (qy < 0 || qy >= Hd || qx < 0 || qx >= Wd) continue;
      float lo = INFINITY;
      for (int ry = qy - 1; ry <= qy + 1; ++ry)
        for (int rx = qx - 1; rx <= qx + 1; ++rx) {
          if (ry < 0 || ry >= Hd || rx < 0 || rx >= Wd) continue;
          const int sy = min((int)floor(ry * ify), Hs - 1), sx = min((int)floor(rx * ifx), Ws - 1);
          lo = fminf(lo, m[(size_t)sy * Ws + sx]);
        }
      best = fmaxf(best, lo);
    }
  dst[b * bstride + (int64_t)dy * Wd + dx] = best;
}

extern "C" int dmh_resize_bilinear_u8(const unsigned char* src, float* dst, int B, int Hs, int Ws, int C, int Hd, int Wd,
                                      int64_t dst_bstride, float div, void* stream) {
  DMH_REQUIRE(src && dst && B > 0 && Hs > 0 && Ws > 0 && C > 0 && Hd > 0 && Wd > 0 && div != 0.f,
              "dmh_resize_bilinear_u8: bad arguments");
  const int64_t total = (int64_t)B * Hd * Wd;
  hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     dst, B, Hs, Ws, C, Hd, Wd, dst_bstride, div);
  DMH_CHECK_LAUNCH("dmh_resize_bilinear_u8");
  return DMH_OK;
}

extern "C" int dmh_mask_open_nearest(const float* src, float* dst, int B, int Hs, int Ws, int Hd, int Wd,
                                     int64_t dst_bstride, void* stream) {
  DMH_REQUIRE(src && dst && B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "dmh_mask_open_nearest: bad arguments");
  const int64_t total = (int64_t)B * Hd * Wd;
  hipLaunchKernelGGL(mask_open_nearest_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     B, Hs, Ws, Hd, Wd, dst_bstride);
  DMH_CHECK_LAUNCH("dmh_mask_open_nearest");
  return DMH_OK;
}
