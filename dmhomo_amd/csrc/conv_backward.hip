// Training: weight and bias gradient of the stride-1 convolutions.
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
// KH: taps per axis; NCB: 16-channel input blocks per workgroup (4 = 64 channels; 1 for the 7x7 init conv, whose 49 taps
// fill the accumulator file on their own); PAD: zero padding of the conv (KH/2 for the 'same' convs, 0 for the 2x2 conv
// over the space-to-depth view that stands for the 4x4 / stride-2 Downsample)
template <int KH, int NCB, int PAD>
struct WgCfg {
  static constexpr int NT = KH * KH;
  static constexpr int XH = TH + KH - 1, XW = TW + KH - 1, XPIX = XH * XW;
  static constexpr int DPX = NCB == 4 ? 80 : 48;            // X pitch: NCB*16 channels + padding, == 16 mod 32
  static constexpr int DY_FLOATS = TP * DP;
  static constexpr int X_FLOATS = XPIX * DPX;
  static constexpr int LDS_BYTES = (DY_FLOATS + X_FLOATS) * 4;
  static constexpr int NDY = TP * 16 / 256;                 // float4 staging slots per thread: dY tile
  static constexpr int XQ = NCB * 4;                        // channel quads per X pixel
  static constexpr int NX = (XPIX * XQ + 255) / 256;        //                                  X halo
};
}  // namespace

struct WgArgs {
  const float* dy;       // [B][H][W][Cout]
  const float* src0;     // [B][H][W][C0]
  const float* src1;     // [B][H][W][C1] or null
  const float* in_coef;  // [B][2][C0] or null: X = SiLU(a * src0 + b) (the consumer-side GroupNorm prologue of dmh_conv2d)
  float* part_w;         // [nsplit][npairs][64 o][64 c][taps]
  float* part_b;         // [nsplit][otiles][64]
  int B, H, W, C0, C1, Cout;  // H, W: size of dy (= conv output)
  int Hin, Win, ups;          // stored input size; ups: the conv saw the nearest-x2 upsampling of it (Upsample, CFG:106-107)
  int tilesX, tilesY, nitems, nsplit, ctiles;
};

template <int KH, int NCB, int PAD>
__global__ __launch_bounds__(256, NCB == 4 ? 2 : 1) void conv_wgrad_kernel(WgArgs p) {
  using Cfg = WgCfg<KH, NCB, PAD>;
  constexpr int NT = Cfg::NT, XW = Cfg::XW, XPIX = Cfg::XPIX, DPX = Cfg::DPX, XQ = Cfg::XQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dyt = lds;
  float* xt = lds + Cfg::DY_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int split = blockIdx.x;
  const int pair = blockIdx.y, ot = pair / p.ctiles, ct = pair % p.ctiles;
  const int o0 = ot * 64, c0 = ct * (NCB * 16);
  const int Cin = p.C0 + p.C1;

  float4v acc[NT][NCB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = float4v{0.f, 0.f, 0.f, 0.f};
  const int Hv = p.ups ? p.Hin * 2 : p.Hin, Wv = p.ups ? p.Win * 2 : p.Win;  // size of the input as the conv saw it
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
    // ---- X halo: (TH + KH - 1) x (TW + KH - 1) pixels x NCB*16 input channels of the concatenated, activated input
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i) {
      const int slot = tid + 256 * i;
      const int pix = slot / XQ, qx = slot % XQ;
      if (pix < XPIX) {
        const int y = oy0 - PAD + pix / XW, x = ox0 - PAD + pix % XW, c = c0 + qx * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < Hv && x >= 0 && x < Wv && c < Cin) {
          const int sy = p.ups ? (y >> 1) : y, sx = p.ups ? (x >> 1) : x;
          const size_t pixoff = (size_t)(b * p.Hin + sy) * p.Win + sx;
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
        st4(xt + pix * DPX + qx * 4, v);
      }
    }
    __syncthreads();
    // ---- 16 K steps of 4 pixels (one tile row quarter each): A = dY, B = X shifted by the tap
#pragma unroll 2
    for (int q = 0; q < TP / 4; ++q) {
      const int pix = q * 4 + kg;  // the K slot of this lane
      const float a = dyt[pix * DP + wave * 16 + l15];
      bsum += a;
      const float* xb = xt + ((pix / TW) * XW + pix % TW) * DPX + l15;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[((t / KH) * XW + t % KH) * DPX + cb * 16], acc[t][cb], 0,
                                                            0, 0);
    }
  }

  // ---- partial block: C/D layout lane (col = c = l15, rows = o = 4*kg + r)
  float* pw = p.part_w + ((size_t)(split * gridDim.y + pair) * 64 * 64) * NT;  // block of 64 o x 64 c slots (NCB*16 used)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pw[((size_t)(wave * 16 + 4 * kg + r) * 64 + cb * 16 + l15) * NT + t] = acc[t][cb][r];
  if (ct == 0) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (kg == 0) p.part_b[((size_t)split * (gridDim.y / p.ctiles) + ot) * 64 + wave * 16 + l15] = bsum;
  }
}

// dW[o][c][tap] (OIHW, Cin = C0 + C1) = sum over the pixel splits; same for db.  A workgroup owns 64 consecutive outputs
// (one 256-byte row of every partial block: coalesced) and spreads the splits over its 4 waves: wave g adds splits
// g, g+4, ... in order, then the four sums are combined as (s0 + s1) + (s2 + s3) — a fixed tree, the same bits every run.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part_w,
                                                                const float* __restrict__ part_b, float* __restrict__ dw,
                                                                float* __restrict__ db, int Cout, int Cin, int NT,
                                                                int nsplit, int otiles, int ctiles, int cw) {
  __shared__ float red[4][64];
  const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + j;
  const int64_t total = (int64_t)Cout * Cin * NT;
  float s = 0.f;
  if (idx < total) {
    const int t = idx % NT;
    const int c = (idx / NT) % Cin;
    const int o = idx / ((int64_t)NT * Cin);
    const int pair = (o / 64) * ctiles + c / cw;  // cw input channels per workgroup block (64, or 16 for the 7x7 conv)
    const size_t off = ((size_t)pair * 64 * 64 + (size_t)(o % 64) * 64 + c % cw) * NT + t;
    const size_t stride = (size_t)otiles * ctiles * 64 * 64 * NT;
    const float* src = part_w + off;
    int sp = g;
    for (; sp + 12 < nsplit; sp += 16) {  // four loads in flight
      const float v0 = src[(size_t)sp * stride], v1 = src[(size_t)(sp + 4) * stride];
      const float v2 = src[(size_t)(sp + 8) * stride], v3 = src[(size_t)(sp + 12) * stride];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; sp < nsplit; sp += 4) s += src[(size_t)sp * stride];
  }
  red[g][j] = s;
  __syncthreads();
  if (g == 0 && idx < total) dw[idx] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
  if (db && blockIdx.x == 0) {
    for (int o = threadIdx.x; o < Cout; o += 256) {
      float sb = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) sb += part_b[((size_t)sp * otiles + o / 64) * 64 + o % 64];
      db[o] = sb;
    }
  }
}

// pixel splits: about 512 workgroups per launch (2 per CU, the kernel's occupancy): every extra split costs a 64x64xtaps
// partial block written and read back
static int wgrad_splits(int nitems, int npairs) {
  int s = 512 / (npairs > 0 ? npairs : 1);
  if (s < 1) s = 1;
  if (s > nitems) s = nitems;
  return s;
}
static int wgrad_cw(int KH) { return KH == 7 ? 16 : 64; }  // input channels per workgroup block

// H, W: size of dy.  KH = 1, 3, 7: 'same' stride-1 conv (pad KH/2), optionally behind a nearest x2 upsampling of the
// stored input (ups = 1: the stored input is H/2 x W/2).  KH = 2: 'valid' 2x2 conv whose stored input is (H+1) x (W+1)
// (the space-to-depth view of the Downsample conv).
extern "C" int64_t dmh_conv_wgrad_workspace_floats(int B, int H, int W, int C0, int C1, int Cout, int KH) {
  const int npairs = cdiv(Cout, 64) * cdiv(C0 + C1, wgrad_cw(KH));
  const int nitems = B * cdiv(H, TH) * cdiv(W, TW);
  const int ns = wgrad_splits(nitems, npairs);
  return (int64_t)ns * npairs * 64 * 64 * KH * KH + (int64_t)ns * cdiv(Cout, 64) * 64;
}

template <int KH, int NCB, int PAD>
static void launch_wgrad(const WgArgs& a, dim3 grid, hipStream_t st) {
  using Cfg = WgCfg<KH, NCB, PAD>;
  auto kern = conv_wgrad_kernel<KH, NCB, PAD>;
  if (Cfg::LDS_BYTES > 64 * 1024) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
      attr = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, a);
}

// dw: [Cout][C0+C1][KH][KH]; db: [Cout] or null; work: dmh_conv_wgrad_workspace_floats floats
extern "C" int dmh_conv_wgrad(const float* dy, const float* src0, const float* src1, const float* in_coef, float* dw,
                              float* db, float* work, int B, int H, int W, int C0, int C1, int Cout, int KH, int ups,
                              void* stream) {
  DMH_REQUIRE(dy && src0 && dw && work, "dmh_conv_wgrad: null pointer");
  DMH_REQUIRE(B > 0 && H > 0 && W > 0 && C0 > 0 && Cout > 0 && C0 % 4 == 0 && Cout % 4 == 0 && (!src1 || C1 % 4 == 0),
              "dmh_conv_wgrad: bad shape");
  DMH_REQUIRE(KH == 1 || KH == 2 || KH == 3 || KH == 7, "dmh_conv_wgrad: kernel %dx%d not built", KH, KH);
  DMH_REQUIRE(!(in_coef && src1), "dmh_conv_wgrad: the GroupNorm prologue applies to a single source");
  DMH_REQUIRE(!ups || (KH == 3 && H % 2 == 0 && W % 2 == 0), "dmh_conv_wgrad: ups needs a 3x3 conv and an even output size");
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
  a.ups = ups ? 1 : 0;
  a.Hin = KH == 2 ? H + 1 : (ups ? H / 2 : H);
  a.Win = KH == 2 ? W + 1 : (ups ? W / 2 : W);
  a.tilesX = cdiv(W, TW);
  a.tilesY = cdiv(H, TH);
  a.nitems = B * a.tilesX * a.tilesY;
  const int cw = wgrad_cw(KH);
  a.ctiles = cdiv(a.C0 + a.C1, cw);
  const int otiles = cdiv(Cout, 64), npairs = otiles * a.ctiles;
  a.nsplit = wgrad_splits(a.nitems, npairs);
  a.part_w = work;
  a.part_b = work + (int64_t)a.nsplit * npairs * 64 * 64 * KH * KH;
  dim3 grid(a.nsplit, npairs);
  switch (KH) {
    case 1: launch_wgrad<1, 4, 0>(a, grid, st); break;
    case 2: launch_wgrad<2, 4, 0>(a, grid, st); break;
    case 3: launch_wgrad<3, 4, 1>(a, grid, st); break;
    default: launch_wgrad<7, 1, 3>(a, grid, st); break;
  }
  DMH_CHECK_LAUNCH("dmh_conv_wgrad");
  const int64_t total = (int64_t)Cout * (a.C0 + a.C1) * KH * KH;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(256), 0, st,
                     a.part_w, a.part_b, dw, db, Cout, a.C0 + a.C1, KH * KH, a.nsplit, otiles, a.ctiles, cw);
  DMH_CHECK_LAUNCH("dmh_conv_wgrad(reduce)");
  return DMH_OK;
}

// ------------------------------------------------------------------------------------------ layout helpers
// shifted space-to-depth view of x [B][H][W][C] (H, W even): X[b][cy][cx][(py*2+px)*C + c] = x[b][2cy-1+py][2cx-1+px][c]
// (zero outside), cy in [0, H/2], cx in [0, W/2] — the 2x2 / stride-1 form of the 4x4 / stride-2 Downsample conv (see
// conv_f16x3.hip); used to take its weight gradient with the stride-1 kernel above.
__global__ void s2d_shift_kernel(const float* __restrict__ x, float* __restrict__ X, int B, int H, int W, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of X
  const int C4 = C / 4, Hc = H / 2 + 1, Wc = W / 2 + 1;
  const int64_t total = (int64_t)B * Hc * Wc * 4 * C4;
  if (i >= total) return;
  const int q = i % C4;
  int64_t r = i / C4;
  const int par = r % 4;
  r /= 4;
  const int cx = r % Wc;
  r /= Wc;
  const int cy = r % Hc;
  const int b = r / Hc;
  const int y = 2 * cy - 1 + (par >> 1), xx = 2 * cx - 1 + (par & 1);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y >= 0 && y < H && xx >= 0 && xx < W) v = ld4(x + ((size_t)(b * H + y) * W + xx) * C + q * 4);
  st4(X + i * 4, v);
}

// pixel shuffle: out[b][2m+ry][2l+rx][c] = in[b][m][l][(ry*2+rx)*C + c]   (in: [B][H][W][4C] -> out: [B][2H][2W][C])
__global__ void d2s_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of in
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * H * W * 4 * C4;
  if (i >= total) return;
  const int q = i % C4;
  int64_t r = i / C4;
  const int par = r % 4;
  r /= 4;
  const int l = r % W;
  r /= W;
  const int m = r % H;
  const int b = r / H;
  st4(out + ((size_t)(b * 2 * H + 2 * m + (par >> 1)) * 2 * W + 2 * l + (par & 1)) * C + q * 4, ld4(in + i * 4));
}

// 2x2 sum pooling: out[b][m][l][c] = sum of in[b][2m+{0,1}][2l+{0,1}][c]  (backward of the nearest x2 upsampling)
__global__ void sumpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of out ([B][H][W][C])
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * H * W * C4;
  if (i >= total) return;
  const int q = i % C4;
  int64_t r = i / C4;
  const int l = r % W;
  r /= W;
  const int m = r % H;
  const int b = r / H;
  const float* p = in + ((size_t)(b * 2 * H + 2 * m) * 2 * W + 2 * l) * C + q * 4;
  const float4 a = ld4(p), bb = ld4(p + C), c = ld4(p + (size_t)2 * W * C), d = ld4(p + (size_t)2 * W * C + C);
  st4(out + i * 4, make_float4((a.x + bb.x) + (c.x + d.x), (a.y + bb.y) + (c.y + d.y), (a.z + bb.z) + (c.z + d.z),
                               (a.w + bb.w) + (c.w + d.w)));
}

extern "C" int dmh_s2d_shift(const float* x, float* X, int B, int H, int W, int C, void* stream) {
  DMH_REQUIRE(x && X && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "dmh_s2d_shift: bad shape");
  const int64_t total = (int64_t)B * (H / 2 + 1) * (W / 2 + 1) * C;
  hipLaunchKernelGGL(s2d_shift_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, X, B, H, W, C);
  DMH_CHECK_LAUNCH("dmh_s2d_shift");
  return DMH_OK;
}
extern "C" int dmh_d2s(const float* in, float* out, int B, int H, int W, int C, void* stream) {
  DMH_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "dmh_d2s: bad shape");
  const int64_t total = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(d2s_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C);
  DMH_CHECK_LAUNCH("dmh_d2s");
  return DMH_OK;
}
extern "C" int dmh_sumpool2(const float* in, float* out, int B, int H, int W, int C, void* stream) {
  DMH_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C % 4 == 0, "dmh_sumpool2: bad shape");
  const int64_t total = (int64_t)B * H * W * C / 4;
  hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C);
  DMH_CHECK_LAUNCH("dmh_sumpool2");
  return DMH_OK;
}
