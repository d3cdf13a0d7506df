// Training, first pieces (in progress): weight and bias gradient of the stride-1 convolutions.
//
//   dW[o][c][ky][kx] = sum_{b,y,x} dY[b][y][x][o] * X[b][y+ky-p][x+kx-p][c]          db[o] = sum_{b,y,x} dY[b][y][x][o]
//
// with X the convolution's (possibly concatenated, possibly GroupNorm+SiLU-prologued) input exactly as dmh_conv2d saw
// it.  (The data gradient needs no kernel of its own: it is dmh_conv2d with the flipped, transposed weight.)
//
// A GEMM whose K axis is the PIXELS: M = output channels, N = input channels x taps.  On v_mfma_f32_16x16x4_f32 (exact
// fp32) one lane feeds one value per operand — A[m = o][k] = dY[pixel k][o], B[k][n = c] = X[pixel k + tap][c] — so
// both operands are read straight from their NHWC tiles in LDS (lanes along the contiguous channel axis, the four K
// slots along four consecutive pixels), no transposition anywhere.
//   workgroup = 4 waves: a 64 (o) x 64 (c) x taps block of dW; wave w owns output channels 16w..16w+15, all four
//               16-channel c blocks and all taps (36 accumulators of 16x16 for 3x3);
//   it walks a contiguous range of (sample, 4x16-pixel tile) items, staging dY (64 px x 64 o) and the X halo
//   (6x18 px x 64 c for 3x3) per item, and writes ONE partial block at the end; a second kernel adds the partials of
//   the pixel splits in a fixed order (deterministic, no float atomics).
// Replaces (with dmh_conv2d for the data gradient): autograd of F.conv2d at CFG:128 inside loss.backward(), DDP:1852.
#include "common.h"

typedef float float4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int TH = 4, TW = 16, TP = TH * TW;  // pixels per item
constexpr int DP = 80;                        // LDS pitch (floats) of a pixel's 64 channels: 80 mod 32 = 16 keeps the four
                                              // K slots of a wave-wide ds_read_b32 on distinct banks
template <int KH>
struct WgCfg {
  static constexpr int NT = KH * KH;
  static constexpr int XH = TH + KH - 1, XW = TW + KH - 1, XPIX = XH * XW;
  static constexpr int DY_FLOATS = TP * DP;
  static constexpr int X_FLOATS = XPIX * DP;
  static constexpr int LDS_BYTES = (DY_FLOATS + X_FLOATS) * 4;
  static constexpr int NDY = TP * 16 / 256;                 // float4 staging slots per thread: dY tile
  static constexpr int NX = (XPIX * 16 + 255) / 256;        //                                  X halo
};
}  // namespace

struct WgArgs {
  const float* dy;       // [B][H][W][Cout]
  const float* src0;     // [B][H][W][C0]
  const float* src1;     // [B][H][W][C1] or null
  const float* in_coef;  // [B][2][C0] or null: X = SiLU(a * src0 + b) (the consumer-side GroupNorm prologue of dmh_conv2d)
  float* part_w;         // [nsplit][npairs][64 o][64 c][taps]
  float* part_b;         // [nsplit][otiles][64]
  int B, H, W, C0, C1, Cout;
  int tilesX, tilesY, nitems, nsplit, ctiles;
};

template <int KH>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgArgs p) {
  using Cfg = WgCfg<KH>;
  constexpr int NT = Cfg::NT, XW = Cfg::XW, XPIX = Cfg::XPIX, PAD = KH / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dyt = lds;
  float* xt = lds + Cfg::DY_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int split = blockIdx.x;
  const int pair = blockIdx.y, ot = pair / p.ctiles, ct = pair % p.ctiles;
  const int o0 = ot * 64, c0 = ct * 64;
  const int Cin = p.C0 + p.C1;

  float4v acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[t][cb] = float4v{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  // contiguous item range of this split
  const int per = (p.nitems + p.nsplit - 1) / p.nsplit;
  const int i0 = split * per, i1 = min(i0 + per, p.nitems);
  const int q4 = tid & 15;  // channel quad of this thread's staging slots

  for (int item = i0; item < i1; ++item) {
    const int tx = item % p.tilesX, ty = (item / p.tilesX) % p.tilesY, b = item / (p.tilesX * p.tilesY);
    const int oy0 = ty * TH, ox0 = tx * TW;
    __syncthreads();  // the previous item's tiles are consumed
    // ---- dY tile: 64 pixels x 64 output channels (zero outside the image / beyond Cout)
#pragma unroll
    for (int i = 0; i < Cfg::NDY; ++i) {
      const int pix = (tid >> 4) + 16 * i;
      const int y = oy0 + pix / TW, x = ox0 + pix % TW, o = o0 + q4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < p.H && x < p.W && o < p.Cout) v = ld4(p.dy + ((size_t)(b * p.H + y) * p.W + x) * p.Cout + o);
      st4(dyt + pix * DP + q4 * 4, v);
    }
    // ---- X halo: (TH + KH - 1) x (TW + KH - 1) pixels x 64 input channels of the concatenated, activated input
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i) {
      const int pix = (tid >> 4) + 16 * i;
      if (pix < XPIX) {
        const int y = oy0 - PAD + pix / XW, x = ox0 - PAD + pix % XW, c = c0 + q4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < p.H && x >= 0 && x < p.W && c < Cin) {
          const size_t pixoff = (size_t)(b * p.H + y) * p.W + x;
          if (c < p.C0) {
            v = ld4(p.src0 + pixoff * p.C0 + c);
            if (p.in_coef) {
              const float4 a = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + c);
              const float4 bb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + c);
              v.x = silu_f(fmaf(a.x, v.x, bb.x));
              v.y = silu_f(fmaf(a.y, v.y, bb.y));
              v.z = silu_f(fmaf(a.z, v.z, bb.z));
              v.w = silu_f(fmaf(a.w, v.w, bb.w));
            }
          } else {
            v = ld4(p.src1 + pixoff * p.C1 + (c - p.C0));
          }
        }
        st4(xt + pix * DP + q4 * 4, v);
      }
    }
    __syncthreads();
    // ---- 16 K steps of 4 pixels (one tile row quarter each): A = dY, B = X shifted by the tap
#pragma unroll 2
    for (int q = 0; q < TP / 4; ++q) {
      const int pix = q * 4 + kg;  // the K slot of this lane
      const float a = dyt[pix * DP + wave * 16 + l15];
      bsum += a;
      const float* xb = xt + ((pix / TW) * XW + pix % TW) * DP + l15;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[((t / KH) * XW + t % KH) * DP + cb * 16], acc[t][cb], 0,
                                                            0, 0);
    }
  }

  // ---- partial block: C/D layout lane (col = c = l15, rows = o = 4*kg + r)
  float* pw = p.part_w + ((size_t)(split * gridDim.y + pair) * 64 * 64) * NT;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pw[((size_t)(wave * 16 + 4 * kg + r) * 64 + cb * 16 + l15) * NT + t] = acc[t][cb][r];
  if (ct == 0) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (kg == 0) p.part_b[((size_t)split * (gridDim.y / p.ctiles) + ot) * 64 + wave * 16 + l15] = bsum;
  }
}

// dW[o][c][tap] (OIHW, Cin = C0 + C1) = sum over the splits, in split order; same for db
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ part_w, const float* __restrict__ part_b,
                                         float* __restrict__ dw, float* __restrict__ db, int Cout, int Cin, int NT,
                                         int nsplit, int otiles, int ctiles) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Cout * Cin * NT;
  if (idx < total) {
    const int t = idx % NT;
    const int c = (idx / NT) % Cin;
    const int o = idx / ((int64_t)NT * Cin);
    const int pair = (o / 64) * ctiles + c / 64;
    const size_t off = ((size_t)pair * 64 * 64 + (size_t)(o % 64) * 64 + c % 64) * NT + t;
    const size_t stride = (size_t)otiles * ctiles * 64 * 64 * NT;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part_w[sp * stride + off];
    dw[idx] = s;
  }
  if (db && idx < Cout) {
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part_b[((size_t)sp * otiles + idx / 64) * 64 + idx % 64];
    db[idx] = s;
  }
}

static int wgrad_splits(int nitems, int npairs) {
  int s = 1024 / (npairs > 0 ? npairs : 1);
  if (s < 1) s = 1;
  if (s > nitems) s = nitems;
  return s;
}

extern "C" int64_t dmh_conv_wgrad_workspace_floats(int B, int H, int W, int C0, int C1, int Cout, int KH) {
  const int npairs = cdiv(Cout, 64) * cdiv(C0 + C1, 64);
  const int nitems = B * cdiv(H, TH) * cdiv(W, TW);
  const int ns = wgrad_splits(nitems, npairs);
  return (int64_t)ns * npairs * 64 * 64 * KH * KH + (int64_t)ns * cdiv(Cout, 64) * 64;
}

// dw: [Cout][C0+C1][KH][KH]; db: [Cout] or null; work: dmh_conv_wgrad_workspace_floats floats
extern "C" int dmh_conv_wgrad(const float* dy, const float* src0, const float* src1, const float* in_coef, float* dw,
                              float* db, float* work, int B, int H, int W, int C0, int C1, int Cout, int KH,
                              void* stream) {
  DMH_REQUIRE(dy && src0 && dw && work, "dmh_conv_wgrad: null pointer");
  DMH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && Cout > 0 && C0 % 4 == 0 && Cout % 4 == 0 && (!src1 || C1 % 4 == 0),
              "dmh_conv_wgrad: bad shape");
  DMH_REQUIRE(KH == 1 || KH == 3, "dmh_conv_wgrad: kernel %dx%d not built yet (stride-1 1x1 and 3x3 only)", KH, KH);
  DMH_REQUIRE(!(in_coef && src1), "dmh_conv_wgrad: the GroupNorm prologue applies to a single source");
  hipStream_t st = (hipStream_t)stream;
  WgArgs a;
  a.dy = dy;
  a.src0 = src0;
  a.src1 = src1;
  a.in_coef = in_coef;
  a.B = B;
  a.H = H;
  a.W = W;
  a.C0 = C0;
  a.C1 = src1 ? C1 : 0;
  a.Cout = Cout;
  a.tilesX = cdiv(W, TW);
  a.tilesY = cdiv(H, TH);
  a.nitems = B * a.tilesX * a.tilesY;
  a.ctiles = cdiv(a.C0 + a.C1, 64);
  const int otiles = cdiv(Cout, 64), npairs = otiles * a.ctiles;
  a.nsplit = wgrad_splits(a.nitems, npairs);
  a.part_w = work;
  a.part_b = work + (int64_t)a.nsplit * npairs * 64 * 64 * KH * KH;
  dim3 grid(a.nsplit, npairs);
  if (KH == 3) {
    hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), WgCfg<3>::LDS_BYTES, st, a);
  } else {
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), WgCfg<1>::LDS_BYTES, st, a);
  }
  DMH_CHECK_LAUNCH("dmh_conv_wgrad");
  const int64_t total = (int64_t)Cout * (a.C0 + a.C1) * KH * KH;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)cdiv64(total > Cout ? total : Cout, 256)), dim3(256), 0, st,
                     a.part_w, a.part_b, dw, db, Cout, a.C0 + a.C1, KH * KH, a.nsplit, otiles, a.ctiles);
  DMH_CHECK_LAUNCH("dmh_conv_wgrad(reduce)");
  return DMH_OK;
}
