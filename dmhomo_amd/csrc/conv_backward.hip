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
  int ablate;  // diagnostics (DMH_WG_ABL): 1 skip the MFMA phase, 2 skip the global loads, 4 skip the split + LDS writes, 8 skip the partial store
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
    bsum = rows_sum(bsum);
    if (kg == 0) p.part_b[((size_t)split * (gridDim.y / p.ctiles) + ot) * 64 + wave * 16 + l15] = bsum;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same GEMM on the fp16 matrix cores (v_mfma_f32_16x16x32_f16), with the operand split of conv_f16x3.hip: each fp32
// operand is two block-scaled fp16 pieces and a product block takes three MFMAs, accumulated in fp32 — the error sits at
// the fp32-accumulation level at about a third of the exact-fp32 kernel's matrix time.
//
// The fp16 MFMA wants its K values (here: PIXELS) 8-contiguous per lane, but NHWC keeps channels contiguous.  The
// transposition happens while staging: a thread loads 4 consecutive pixels x 4 channels (float4 per pixel, 256 B per
// pixel across 16 lanes), scales, splits, and writes per channel one 8-byte run of 4 pixels into channel-major planes
//     dY:  [o][4 rows x 16 px]  fp16, pitch 144 B            X:  [c][6 rows x 24 px]  fp16, pitch 304 B
// (both pitches an odd number of 16-byte units: the 16 lanes of a ds_read_b128 phase hit 16 distinct bank groups).
// A K step of 32 = 4 tile rows (the lane's K group) x 8 pixels of a half row; the tap's x shift (0..2 pixels = 0..4
// bytes) cannot be an aligned 16-byte read, so a lane reads its 8 pixels + the next 2 once per (ky, channel block) and
// forms the three kx operands in registers (v_alignbit for the odd shift, plain re-indexing for the even one).
// Scales: per workgroup, running maxima of |dY| and |X| over its items (as in conv_f16x3.hip): when a later item raises a
// maximum the accumulators are multiplied by the (power-of-two) ratio, so the scale only shrinks and nothing overflows.
//     d1 = fp16(d*sd)  d2 = fp16(d*sd - d1)      x1 = fp16(x*sx)  x2 = fp16(x*sx - x1)      (dmh_split2: six instructions per pair)
//     d*x*sd*sx = d1*x1 + d2*x1 + d1*x2 + O(2^-22)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

namespace {
template <int KH, int NCBW>
struct WhCfg {
  static constexpr int NT = KH * KH;
  static constexpr int XH = TH + KH - 1, XW = TW + KH - 1;
  static constexpr int XG = (XW + 3) / 4;                       // 4-pixel staging groups per halo row
  static constexpr int XROWB = ((XW + 7) / 8 * 8) * 2;          // bytes per halo row (pixels padded to 8)
  static constexpr int XPITCH = ((XH * XROWB / 16) | 1) * 16;   // bytes per input channel: an odd number of 16-byte units
  static constexpr int APITCH = TP * 2 + 16;                    // bytes per output channel: 64 px
  static constexpr int A_BYTES = 64 * APITCH, X_BYTES = 16 * NCBW * XPITCH;
  static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * X_BYTES + 64;   // planes + the maxima slots
  static constexpr int XQN = 4 * NCBW;                          // channel quads of the workgroup's input channels
  static constexpr int XSLOTS = XH * XG * XQN;                  // X staging slots (4 px x 4 c each) per item, <= 256
  static constexpr int OBG = 4 / NCBW;                          // wave -> (output-block group, local channel block)
};
__device__ __forceinline__ float amax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ int max_exponent(unsigned bits) {  // v in [2^e, 2^(e+1)); all-zero / tiny tiles clamp at -100
  const int e = (int)((bits >> 23) & 0xff) - 127;
  return e < -100 ? -100 : e;
}
}  // namespace

// Workgroup = 4 waves: 64 output channels (wave w: 16w..16w+15) x ONE 16-channel input block x all taps, over a
// contiguous range of (sample, 4x16-pixel tile) items.  grid = (pixel splits, 64x64 pairs * 4 channel blocks): a
// quarter of the accumulators of the fp32 kernel (9 x 4 registers for 3x3), so four workgroups fit a CU and hide each
// other's staging, and a pixel split costs a quarter of the partial block.  dY is staged by all four channel-block
// workgroups of a pair (they share a split index modulo 8, hence an XCD and its L2).
template <int KH, int PAD, int NCBW>
__global__ __launch_bounds__(256, NCBW == 1 ? 3 : 2) void conv_wgrad_f16x3_kernel(WgArgs p) {
  using Cfg = WhCfg<KH, NCBW>;
  constexpr int OBW = NCBW, OBG = Cfg::OBG;  // output blocks per wave; waves per channel block
  constexpr int NT = Cfg::NT, XW = Cfg::XW, XG = Cfg::XG, XROWB = Cfg::XROWB, XPITCH = Cfg::XPITCH, APITCH = Cfg::APITCH;
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  char* a1 = ldsb;
  char* a2 = a1 + Cfg::A_BYTES;
  char* x1 = a2 + Cfg::A_BYTES;
  char* x2 = x1 + Cfg::X_BYTES;
  unsigned* mx = reinterpret_cast<unsigned*>(x2 + Cfg::X_BYTES);  // [parity][0 dY, 1 X] max |value| bits

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const int split = blockIdx.x;
  const int pair = blockIdx.y / OBG, cbq = blockIdx.y % OBG, ot = pair / p.ctiles, ct = pair % p.ctiles;
  const int o0 = ot * 64, c0 = ct * 64 + cbq * 16 * NCBW;
  const int obg = wave % OBG, cbl = wave / OBG;  // this wave: output blocks obg*OBW .. +OBW-1, local channel block cbl
  const int Cin = p.C0 + p.C1;
  const int Hv = p.ups ? p.Hin * 2 : p.Hin, Wv = p.ups ? p.Win * 2 : p.Win;
  if (c0 >= Cin) return;  // a channel block beyond the input channels (Cin not a multiple of 64): its slice stays unread

  float4v acc[OBW][NT];
#pragma unroll
  for (int i = 0; i < OBW; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = float4v{0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);  // this thread's 4 output channels, summed over its pixels
  int ed = -127, ex = -127;                        // running exponents of the maxima
  if (tid < 4) mx[tid] = 0u;
  __syncthreads();

  const int per = (p.nitems + p.nsplit - 1) / p.nsplit;
  const int i0 = split * per, i1 = min(i0 + per, p.nitems);
  const int q4 = tid & 15, pg = tid >> 4;  // dY staging: channel quad, pixel group
  const int xq = tid % Cfg::XQN, xgi = tid / Cfg::XQN;  // X staging: channel quad of this workgroup's, (row, pixel group)
  const int xrow = xgi / XG, xg = xgi % XG;
  const bool xslot = tid < Cfg::XSLOTS;

  // Staging registers of one item: raw loads (issued from clamped addresses, so that they can fly during the previous
  // item's MFMA phase — hipcc drains vmcnt around loads that sit behind a branch) + validity bits applied on use.
  float4 dv[4], xv[4], pa, pb;
  unsigned vmask = 0u;  // bit j: dv[j] valid; bit 4+j: xv[j] valid; bit 8: xv takes the GroupNorm+SiLU prologue
  auto issue_loads = [&](int item) {
    const int tx = item % p.tilesX, ty = (item / p.tilesX) % p.tilesY, b = item / (p.tilesX * p.tilesY);
    const int oy0 = ty * TH, ox0 = tx * TW;
    unsigned vm = 0u;
    {
      const int y = oy0 + (pg >> 2), o = o0 + q4 * 4;
      const int yc = min(y, p.H - 1), oc = min(o, p.Cout - 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = ox0 + (pg & 3) * 4 + j;
        if (y < p.H && x < p.W && o < p.Cout) vm |= 1u << j;
        dv[j] = ld4(p.dy + ((size_t)(b * p.H + yc) * p.W + min(x, p.W - 1)) * p.Cout + oc);
      }
    }
    {
      const int c = c0 + xq * 4, cc = min(c, Cin - 4);
      const int y = oy0 - PAD + xrow, yc = min(max(y, 0), Hv - 1);
      const int sy = p.ups ? (yc >> 1) : yc;
      const bool first = cc < p.C0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xl = xg * 4 + j, x = ox0 - PAD + xl;
        if (xslot && xl < XW && y >= 0 && y < Hv && x >= 0 && x < Wv && c < Cin) vm |= 16u << j;
        const int xc = min(max(x, 0), Wv - 1), sx = p.ups ? (xc >> 1) : xc;
        const size_t pixoff = (size_t)(b * p.Hin + sy) * p.Win + sx;
        const float* src = first ? p.src0 + pixoff * p.C0 + cc : p.src1 + pixoff * p.C1 + (cc - p.C0);
        xv[j] = ld4(src);
      }
      const int cp = min(cc, p.C0 - 4);
      const float* cf = p.in_coef ? p.in_coef + (size_t)(b * 2) * p.C0 + cp : p.src0;
      pa = ld4(cf);
      pb = ld4(cf + (p.in_coef ? p.C0 : 0));
      if (p.in_coef && first) vm |= 256u;
    }
    vmask = vm;
  };
  if (i0 < i1) issue_loads(i0);

  for (int item = i0; item < i1; ++item) {
    const int par = (item - i0) & 1;
    // ---- the staged registers become values: zero outside the image / channels, the consumer-side prologue
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!(vmask & (1u << j)) || (p.ablate & 2)) dv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v = xv[j];
      if (vmask & 256u) {
        v.x = silu_f(fmaf(pa.x, v.x, pb.x));
        v.y = silu_f(fmaf(pa.y, v.y, pb.y));
        v.z = silu_f(fmaf(pa.z, v.z, pb.z));
        v.w = silu_f(fmaf(pa.w, v.w, pb.w));
      }
      if (!(vmask & (16u << j)) || (p.ablate & 2)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      xv[j] = v;
    }
    // ---- tile maxima -> LDS (non-negative floats order like their bit patterns)
    float md = 0.f, mxx = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      md = fmaxf(md, amax4(dv[j]));
      mxx = fmaxf(mxx, amax4(xv[j]));
      bsum.x += dv[j].x;
      bsum.y += dv[j].y;
      bsum.z += dv[j].z;
      bsum.w += dv[j].w;
    }
    md = __uint_as_float(wave_max_u32(__float_as_uint(md)));     // (magnitudes: their bit patterns order like the values)
    mxx = __uint_as_float(wave_max_u32(__float_as_uint(mxx)));
    if (lane == 0) {
      atomicMax(&mx[par * 2 + 0], __float_as_uint(md));
      atomicMax(&mx[par * 2 + 1], __float_as_uint(mxx));
    }
    __syncthreads();  // (A) maxima complete; every wave is past the previous item's fragment reads
    const int nd = max(ed, max_exponent(mx[par * 2 + 0])), nx = max(ex, max_exponent(mx[par * 2 + 1]));
    if (nd + nx != ed + ex) {  // workgroup-uniform
      const float f = ldexpf(1.f, max((ed + ex) - (nd + nx), -120));
#pragma unroll
      for (int i = 0; i < OBW; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] *= f;
      ed = nd;
      ex = nx;
    }
    if (tid < 2) mx[(par ^ 1) * 2 + tid] = 0u;  // the other parity's slots, for the next item
    const float sd = ldexpf(1.f, 14 - ed), sx_ = ldexpf(1.f, 14 - ex);
    // ---- split + transposed LDS writes: per channel one run of 4 pixels
    if (!(p.ablate & 4)) {
      const int pixb = ((pg >> 2) * TW + (pg & 3) * 4) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = i == 0 ? dv[j].x : i == 1 ? dv[j].y : i == 2 ? dv[j].z : dv[j].w;
        uint2 h1, h2;
        dmh_split2(e[0], e[1], sd, h1.x, h2.x);
        dmh_split2(e[2], e[3], sd, h1.y, h2.y);
        *reinterpret_cast<uint2*>(a1 + (q4 * 4 + i) * APITCH + pixb) = h1;
        *reinterpret_cast<uint2*>(a2 + (q4 * 4 + i) * APITCH + pixb) = h2;
      }
      if (xslot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = i == 0 ? xv[j].x : i == 1 ? xv[j].y : i == 2 ? xv[j].z : xv[j].w;
          uint2 h1, h2;   // (second piece unscaled, as everywhere since round 2: conv_f16x3.hip)
          dmh_split2(e[0], e[1], sx_, h1.x, h2.x);
          dmh_split2(e[2], e[3], sx_, h1.y, h2.y);
          *reinterpret_cast<uint2*>(x1 + (xq * 4 + i) * XPITCH + xrow * XROWB + xg * 8) = h1;
          *reinterpret_cast<uint2*>(x2 + (xq * 4 + i) * XPITCH + xrow * XROWB + xg * 8) = h2;
        }
      }
    }
    __syncthreads();  // (B) tiles staged
    issue_loads(min(item + 1, i1 - 1));  // the next item's pixels fly during this item's matrix phase
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2 K steps of 32 pixels (4 tile rows x 8 pixels of a half row)
    if (!(p.ablate & 1))
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        half8 d1[OBW], d2[OBW];
#pragma unroll
        for (int i = 0; i < OBW; ++i) {
          const char* ap = a1 + ((obg * OBW + i) * 16 + l15) * APITCH + (kg * TW + half * 8) * 2;
          d1[i] = *reinterpret_cast<const half8*>(ap);
          d2[i] = *reinterpret_cast<const half8*>(ap + Cfg::A_BYTES);
        }
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
          const char* xp = x1 + (cbl * 16 + l15) * XPITCH + (kg + ky) * XROWB + half * 16;
          const uint4v u1 = *reinterpret_cast<const uint4v*>(xp);
          const uint4v u2 = *reinterpret_cast<const uint4v*>(xp + Cfg::X_BYTES);
          unsigned e1 = 0u, e2 = 0u;
          if (KH > 1) {
            e1 = *reinterpret_cast<const unsigned*>(xp + 16);
            e2 = *reinterpret_cast<const unsigned*>(xp + Cfg::X_BYTES + 16);
          }
          half8 b1[KH], b2[KH];
#pragma unroll
          for (int kx = 0; kx < KH; ++kx) {
            uint4v s1 = u1, s2 = u2;
            if (kx == 1) {
              s1 = uint4v{__builtin_amdgcn_alignbit(u1.y, u1.x, 16), __builtin_amdgcn_alignbit(u1.z, u1.y, 16),
                          __builtin_amdgcn_alignbit(u1.w, u1.z, 16), __builtin_amdgcn_alignbit(e1, u1.w, 16)};
              s2 = uint4v{__builtin_amdgcn_alignbit(u2.y, u2.x, 16), __builtin_amdgcn_alignbit(u2.z, u2.y, 16),
                          __builtin_amdgcn_alignbit(u2.w, u2.z, 16), __builtin_amdgcn_alignbit(e2, u2.w, 16)};
            } else if (kx == 2) {
              s1 = uint4v{u1.y, u1.z, u1.w, e1};
              s2 = uint4v{u2.y, u2.z, u2.w, e2};
            }
            b1[kx] = __builtin_bit_cast(half8, s1);
            b2[kx] = __builtin_bit_cast(half8, s2);
          }
          // the three terms, each across the kx accumulators: consecutive MFMAs never share an accumulator
#pragma unroll
          for (int i = 0; i < OBW; ++i)
#pragma unroll
            for (int kx = 0; kx < KH; ++kx)
              acc[i][ky * KH + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1[i], b2[kx], acc[i][ky * KH + kx], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < OBW; ++i)
#pragma unroll
            for (int kx = 0; kx < KH; ++kx)
              acc[i][ky * KH + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2[i], b1[kx], acc[i][ky * KH + kx], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < OBW; ++i)
#pragma unroll
            for (int kx = 0; kx < KH; ++kx)
              acc[i][ky * KH + kx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1[i], b1[kx], acc[i][ky * KH + kx], 0, 0, 0);
        }
      }
  }

  // ---- partial block, un-scaled.  Layout [tap][channel block][wave][kg][l15][r]: one 16-byte store per lane, 1 KB
  // contiguous per wave (conv_wgrad_reduce_v2_kernel undoes the permutation while it writes the 147 KB of dW)
  const float osc = ldexpf(1.f, max(ed + ex - 28, -126));
  float* pw = p.part_w + ((size_t)(split * (gridDim.y / OBG) + pair) * 64 * 64) * NT;
  if (!(p.ablate & 8))
#pragma unroll
    for (int i = 0; i < OBW; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float4v v = acc[i][t] * osc;
        *reinterpret_cast<float4v*>(pw + ((size_t)((t * 4 + cbq * NCBW + cbl) * 4 + obg * OBW + i) * 4 + kg) * 64 + l15 * 4) = v;
      }
  if (ct == 0 && cbq == 0) {  // bias partial: the 16 pixel groups of a channel quad -> one sum per output channel, fixed order
    __syncthreads();
    float* br = reinterpret_cast<float*>(a1);  // [16 pg][64 o]
    st4(br + pg * 64 + q4 * 4, bsum);
    __syncthreads();
    if (tid < 64) {
      float sb = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) sb += br[g * 64 + tid];
      p.part_b[((size_t)split * ((gridDim.y / OBG) / p.ctiles) + ot) * 64 + tid] = sb;
    }
  }
}

// dW from the partial blocks of conv_wgrad_f16x3_kernel.  A thread owns one 16-byte unit (4 output channels) of the
// block layout; a workgroup 16 consecutive units x 16 split groups: group g adds splits g, g+16, ... in order, then the
// 16 sums are combined by a fixed tree; the result is scattered to OIHW.  db as in the fp32 path.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_v2_kernel(const float* __restrict__ part_w,
                                                                   const float* __restrict__ part_b, float* __restrict__ dw,
                                                                   float* __restrict__ db, int Cout, int Cin, int NT,
                                                                   int nsplit, int otiles, int ctiles) {
  __shared__ float4 red[16][17];
  const int j = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int64_t unit = (int64_t)blockIdx.x * 16 + j;  // float4 index inside [pair][t][cb][wave][kg][l15]
  const int64_t units = (int64_t)otiles * ctiles * NT * 1024;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = (size_t)otiles * ctiles * 64 * 64 * NT;
  if (unit < units) {
    const float* src = part_w + unit * 4;
    int sp = g;
    for (; sp + 48 < nsplit; sp += 64) {
      const float4 v0 = ld4(src + (size_t)sp * stride), v1 = ld4(src + (size_t)(sp + 16) * stride);
      const float4 v2 = ld4(src + (size_t)(sp + 32) * stride), v3 = ld4(src + (size_t)(sp + 48) * stride);
      s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x;
      s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
      s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z;
      s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; sp < nsplit; sp += 16) {
      const float4 v = ld4(src + (size_t)sp * stride);
      s.x += v.x;
      s.y += v.y;
      s.z += v.z;
      s.w += v.w;
    }
  }
  red[g][j] = s;
  __syncthreads();
  if (g == 0 && unit < units) {
    float4 t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = red[k][j];
#pragma unroll
    for (int w = 8; w; w >>= 1)
#pragma unroll
      for (int k = 0; k < w; ++k) {
        t[k].x = t[2 * k].x + t[2 * k + 1].x;
        t[k].y = t[2 * k].y + t[2 * k + 1].y;
        t[k].z = t[2 * k].z + t[2 * k + 1].z;
        t[k].w = t[2 * k].w + t[2 * k + 1].w;
      }
    int64_t u = unit;
    const int l15 = u % 16;
    u /= 16;
    const int kg = u % 4;
    u /= 4;
    const int wave = u % 4;
    u /= 4;
    const int cb = u % 4;
    u /= 4;
    const int tap = u % NT;
    const int pair = u / NT;
    const int c = (pair % ctiles) * 64 + cb * 16 + l15;
    const int ob = (pair / ctiles) * 64 + wave * 16 + kg * 4;
    const float r[4] = {t[0].x, t[0].y, t[0].z, t[0].w};
    if (c < Cin)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ob + i < Cout) dw[((size_t)(ob + i) * Cin + c) * NT + tap] = r[i];
  }
  // db: workgroup i owns output channels 16i..16i+15, the same 16 split groups and tree
  if (db && (int)blockIdx.x * 16 < Cout) {
    const int o = blockIdx.x * 16 + j;
    float sb = 0.f;
    if (o < Cout)
      for (int sp = g; sp < nsplit; sp += 16) sb += part_b[((size_t)sp * otiles + o / 64) * 64 + o % 64];
    __syncthreads();
    red[g][j].x = sb;
    __syncthreads();
    if (g == 0 && o < Cout) {
      float t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = red[k][j].x;
#pragma unroll
      for (int w = 8; w; w >>= 1)
#pragma unroll
        for (int k = 0; k < w; ++k) t[k] = t[2 * k] + t[2 * k + 1];
      db[o] = t[0];
    }
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
  // db: workgroup i owns output channels 64i..64i+63, the same 4 split groups and tree
  if (db && (int)blockIdx.x * 64 < Cout) {
    const int o = blockIdx.x * 64 + j;
    float sb = 0.f;
    if (o < Cout)
      for (int sp = g; sp < nsplit; sp += 4) sb += part_b[((size_t)sp * otiles + o / 64) * 64 + o % 64];
    __syncthreads();
    red[g][j] = sb;
    __syncthreads();
    if (g == 0 && o < Cout) db[o] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
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
// fp16-piece kernels: about 1024 workgroups (4 per CU) over (splits, pairs * 4 channel blocks); splits a multiple of 8 where
// possible so that the four channel-block workgroups of a pair share an XCD (linear workgroup id modulo 8)
static int wgrad_ncbw() {  // input-channel blocks per workgroup of the fp16-piece kernels (DMH_WGRAD_NCBW=1|2)
  static const int v = [] {
    const char* e = getenv("DMH_WGRAD_NCBW");
    const int k = e ? atoi(e) : 2;
    return k == 1 ? 1 : 2;
  }();
  return v;
}
static int wgrad_splits_f16(int nitems, int npairs) {
  const int ncbw = wgrad_ncbw();
  int s = (ncbw == 1 ? 1024 : 512) / ((4 / ncbw) * (npairs > 0 ? npairs : 1));
  if (s >= 8) s = s / 8 * 8;
  if (s < 1) s = 1;
  if (s > nitems) s = nitems;
  return s;
}
static int wgrad_cw(int KH) { return KH == 7 ? 16 : 64; }  // input channels per workgroup block

// H, W: size of dy.  KH = 1, 3, 7: 'same' stride-1 conv (pad KH/2), optionally behind a nearest x2 upsampling of the
// stored input (ups = 1: the stored input is H/2 x W/2).  KH = 2: 'valid' 2x2 conv whose stored input is (H+1) x (W+1)
// (the space-to-depth view of the Downsample conv).
extern "C" int64_t dmh_conv_wgrad_workspace_floats(int B, int H, int W, int C0, int C1, int Cout, int KH) {
  if (!dmh_dims_ok({B, H, W, C0, Cout}) || !dmh_dims_ok({C1}, 0) || !dmh_dims_ok({KH}, 1, 16) ||
      (long long)B * H * W > (1ll << 31)) return -1;
  const int npairs = cdiv(Cout, 64) * cdiv(C0 + C1, wgrad_cw(KH));
  const int nitems = B * cdiv(H, TH) * cdiv(W, TW);
  int ns = wgrad_splits(nitems, npairs);
  if (KH <= 3 && wgrad_splits_f16(nitems, npairs) > ns) ns = wgrad_splits_f16(nitems, npairs);
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

template <int KH, int PAD, int NCBW>
static void launch_wgrad_f16(const WgArgs& a, dim3 grid, hipStream_t st) {
  using Cfg = WhCfg<KH, NCBW>;
  auto kern = conv_wgrad_f16x3_kernel<KH, PAD, NCBW>;
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
  static const int abl = [] {
    const char* e = getenv("DMH_WG_ABL");
    return e ? atoi(e) : 0;
  }();
  a.ablate = abl;
  static const int variant = [] {  // DMH_WGRAD_VARIANT=0: the exact-fp32 MFMA kernels everywhere
    const char* e = getenv("DMH_WGRAD_VARIANT");
    return e ? atoi(e) : 1;
  }();
  static const int f16_1x1 = [] {
    const char* e = getenv("DMH_WGRAD_F16_1X1");
    return e ? atoi(e) : 0;
  }();
  const bool f16 = variant == 1 && (KH == 2 || KH == 3 || (KH == 1 && f16_1x1));  // 1x1: too little matrix work per staged dY tile, the fp32 kernel wins
  a.nsplit = f16 ? wgrad_splits_f16(a.nitems, npairs) : wgrad_splits(a.nitems, npairs);
  a.part_w = work;
  a.part_b = work + (int64_t)a.nsplit * npairs * 64 * 64 * KH * KH;
  if (f16) {
    if (wgrad_ncbw() == 1) {
      dim3 grid(a.nsplit, npairs * 4);
      if (KH == 2) launch_wgrad_f16<2, 0, 1>(a, grid, st); else launch_wgrad_f16<3, 1, 1>(a, grid, st);
    } else {
      dim3 grid(a.nsplit, npairs * 2);
      if (KH == 1) launch_wgrad_f16<1, 0, 2>(a, grid, st);
      else if (KH == 2) launch_wgrad_f16<2, 0, 2>(a, grid, st);
      else launch_wgrad_f16<3, 1, 2>(a, grid, st);
    }
    DMH_CHECK_LAUNCH("dmh_conv_wgrad");
    const int64_t units = (int64_t)npairs * KH * KH * 1024;
    hipLaunchKernelGGL(conv_wgrad_reduce_v2_kernel, dim3((unsigned)cdiv64(units, 16)), dim3(256), 0, st, a.part_w, a.part_b, dw,
                       db, Cout, a.C0 + a.C1, KH * KH, a.nsplit, otiles, a.ctiles);
  } else {
    dim3 grid(a.nsplit, npairs);
    switch (KH) {
      case 1: launch_wgrad<1, 4, 0>(a, grid, st); break;
      case 2: launch_wgrad<2, 4, 0>(a, grid, st); break;
      case 3: launch_wgrad<3, 4, 1>(a, grid, st); break;
      default: launch_wgrad<7, 1, 3>(a, grid, st); break;
    }
    DMH_CHECK_LAUNCH("dmh_conv_wgrad");
    const int64_t total = (int64_t)Cout * (a.C0 + a.C1) * KH * KH;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)cdiv64(total, 64)), dim3(256), 0, st, a.part_w, a.part_b, dw,
                       db, Cout, a.C0 + a.C1, KH * KH, a.nsplit, otiles, a.ctiles, cw);
  }
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
