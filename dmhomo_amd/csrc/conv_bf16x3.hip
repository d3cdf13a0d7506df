// K1b — convolution as an implicit GEMM on the bf16 matrix cores, carrying fp32 operands as three bf16
// pieces each ("bf16x3 split", six cross terms, fp32 accumulation).
//
// Why: on gfx950 the fp32 MFMAs execute on the vector ALUs (tools/micro/coexec.hip: an fp32-MFMA wave and a
// VALU wave sharing a SIMD take the SUM of their times), so an fp32-MFMA kernel can hide neither its prologue
// nor its epilogue behind the matrix work, and its peak is the packed-fp32 VALU peak (157 TFLOP/s).  The bf16
// MFMAs run on the real matrix cores at 16x that rate and do overlap VALU/LDS work of the other resident waves.
//
// Arithmetic: x = x1 + x2 + x3 with x1 = bf16_trunc(x), x2 = bf16_trunc(x - x1), x3 = bf16_trunc(x - x1 - x2).
// The two subtractions are exact, so the three pieces carry >= 24 significant bits: x is represented EXACTLY
// (for normal x).  Every bf16*bf16 product is exact in fp32; the MFMA accumulates in fp32.  Of the nine cross
// terms the six with i + j <= 4 are kept; the dropped ones (x2*w3, x3*w2, x3*w3) are <= 2^-23 relative to x*w,
// i.e. below the rounding of the fp32 accumulation itself.  Measured against an fp64 convolution the error equals
// that of the fp32-MFMA kernels (tests/test_gpu_kernels.py); the exponent range is that of fp32.
//
//   M = output pixels: each wave owns 64 (two 32-row blocks);  N = 64 output channels per wave (two 32 blocks)
//   workgroup = 4 waves as WM x WN:  4x1 -> 16x16 pixels x 64 cout,   2x2 -> 8x16 pixels x 128 cout
//   K = taps x input channels, walked in chunks of 32 channels; per chunk the (halo) input tile is staged ONCE
//       into LDS — through the fused prologue SiLU(a*x+b) when a GroupNorm is pending — already split into its
//       three bf16 planes, and reused by all taps.  Weights are split at pack time and stored fragment-major, so
//       each B fragment is one coalesced 1 KB global (L2-resident) load per wave, prefetched one K-step ahead.
// Epilogue: shared with conv.hip (LDS transpose, + bias, + residual, float4 NHWC stores, GroupNorm partials).
//
// Replaces: the 3x3 convolutions of Block / ResnetBlock CFG:128-170 and the Upsample conv CFG:106-107.
#include <stdlib.h>

#include "common.h"

#include "conv_args.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KC = 32;        // input channels per chunk
constexpr int PITCH = 208;    // LDS bytes per staged pixel: 3 planes x 32 bf16 + 16 (odd multiple of 16 B: conflict-free b128)
constexpr int FRAG_U4 = 64;   // one B fragment = 64 lanes x 16 B
constexpr int STEP_U4 = 6 * FRAG_U4;  // per (chunk, tap, k16): 2 column blocks x 3 pieces

template <int KH, int KW, int S, int UPS, int TH, int TW, int WM, int WN>
struct SplitCfg {
  static_assert(WM * WN == 4, "four waves per workgroup");
  static_assert(TH * TW == WM * 64, "64 pixels per wave");
  static constexpr int IN_H = (TH - 1) * S + KH;
  static constexpr int IN_W = (TW - 1) * S + KW;
  static constexpr int IN_PIX = IN_H * IN_W;
  static constexpr int NLOAD = (IN_PIX * 8 + 255) / 256;
  static constexpr int PAD = (S == 1) ? (KH / 2) : (KH == 4 ? 1 : 0);
  static constexpr int IN_BYTES = IN_PIX * PITCH;
  static constexpr int EPI_BYTES = 4 * 32 * EpilogueRows::EP * 4;
  static constexpr int LDS_BYTES = IN_BYTES > EPI_BYTES ? IN_BYTES : EPI_BYTES;
};

// top 16 bits of two floats -> one dword of two bf16 (lo = a, hi = b); truncation
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// four floats -> three planes of four bf16 (8 B each)
__device__ __forceinline__ void split4(const float4& x, uint2& p1, uint2& p2, uint2& p3) {
  const float ax = trunc_bf16(x.x), ay = trunc_bf16(x.y), az = trunc_bf16(x.z), aw = trunc_bf16(x.w);
  const float rx = x.x - ax, ry = x.y - ay, rz = x.z - az, rw = x.w - aw;  // exact
  const float bx = trunc_bf16(rx), by = trunc_bf16(ry), bz = trunc_bf16(rz), bw = trunc_bf16(rw);
  const float sx = rx - bx, sy = ry - by, sz = rz - bz, sw = rw - bw;  // exact
  p1 = make_uint2(pack_hi16(x.x, x.y), pack_hi16(x.z, x.w));
  p2 = make_uint2(pack_hi16(rx, ry), pack_hi16(rz, rw));
  p3 = make_uint2(pack_hi16(sx, sy), pack_hi16(sz, sw));
}

}  // namespace

// ABL: compile-time ablation for diagnostic builds (1: no weight loads after the first step, 2: A fragments
// read once per chunk) — results are wrong, only the timing is of interest.  Always 0 in the product library.
template <int KH, int KW, int S, int UPS, int TH, int TW, int WM, int WN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_bf16x3_kernel(ConvArgs p) {
  using Cfg = SplitCfg<KH, KW, S, UPS, TH, TW, WM, WN>;
  constexpr int IN_W = Cfg::IN_W, IN_PIX = Cfg::IN_PIX, NLOAD = Cfg::NLOAD, NTAPS = KH * KW;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned char* in_tile = reinterpret_cast<unsigned char*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  int t = blockIdx.x;
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY;
  const int b = t / p.tilesY;
  const int nt = blockIdx.y * WN + wn;
  const int n0 = nt * 64;

  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - Cfg::PAD, ix0 = ox0 * S - Cfg::PAD;
  const int Hlim = UPS ? p.Hin * 2 : p.Hin;
  const int Wlim = UPS ? p.Win * 2 : p.Win;

  // LDS byte offset of this lane's A rows at tap (0,0), piece 0, k16 step 0
  int arow[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int r = wm * 64 + mb * 32 + l31;
    const int py = r / TW, px = r % TW;
    arow[mb] = ((py * S) * IN_W + px * S) * PITCH + half * 16;
  }

  floatx16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const int nchunks = p.nch0 + p.nch1;
  const int nsteps = nchunks * NTAPS * 2;
  const uint4* wbase = reinterpret_cast<const uint4*>(p.wpack) + (size_t)nt * nsteps * STEP_U4 + lane;

  // B fragments of one K step: [column block][piece]; double buffered across steps
  uint4 bq[2][6];
  auto load_b = [&](int buf, int step) {
    const uint4* src = wbase + (size_t)(step < nsteps ? step : nsteps - 1) * STEP_U4;
#pragma unroll
    for (int i = 0; i < 6; ++i) bq[buf][i] = src[i * FRAG_U4];
  };
  load_b(0, 0);
  if (ABL & 1) load_b(1, 1);
  bf16x8 a[2][3];

  // input (halo) tile of one channel chunk: global -> registers, issued one chunk ahead.  Unconditional loads
  // from clamped addresses + a validity mask (see conv.hip) keep them in flight under counted waits.
  const int c4 = tid & 7;
  float4 v[NLOAD];
  float4 ca, cb;
  int poff[NLOAD];
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int pix = (tid + i * 256) >> 3;
    const int pixc = pix < IN_PIX ? pix : IN_PIX - 1;
    const int yy = iy0 + pixc / IN_W, xx = ix0 + pixc % IN_W;
    const bool ok = pix < IN_PIX && yy >= 0 && yy < Hlim && xx >= 0 && xx < Wlim;
    const int yc = min(max(yy, 0), Hlim - 1), xc = min(max(xx, 0), Wlim - 1);
    const int sy = UPS ? (yc >> 1) : yc, sx = UPS ? (xc >> 1) : xc;
    poff[i] = (b * p.Hin + sy) * p.Win + sx;
    inside |= (ok ? 1u : 0u) << i;
  }
  auto issue_chunk_loads = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    const float* src = s1 ? p.src1 : p.src0;
    const int Csrc = s1 ? p.C1 : p.C0;
    const int c = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4;
    const int cc = c < Csrc ? c : 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = ld4(src + (size_t)poff[i] * Csrc + cc);
    ca = make_float4(1.f, 1.f, 1.f, 1.f);
    cb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.in_coef != nullptr && !s1) {
      ca = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + cc);
      cb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + cc);
    }
  };
  issue_chunk_loads(0);
  const int wr0 = (tid >> 3) * PITCH + c4 * 8;  // staging slot 0; slot i is 32 pixels further

  int step = 0;
  for (int ch = 0; ch < nchunks; ++ch) {
    const bool s1c = ch >= p.nch0;
    const bool pro = (p.in_coef != nullptr) && !s1c;
    const bool cvalid = (s1c ? ch - p.nch0 : ch) * KC + c4 * 4 < (s1c ? p.C1 : p.C0);
    const unsigned msk = cvalid ? inside : 0u;
    __syncthreads();  // every wave is done reading the previous chunk's tile
#ifdef DMH_STAMPS
    if (!(p.ablate & 1))
#endif
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if ((i + 1) * 256 <= IN_PIX * 8 || ((tid + i * 256) >> 3) < IN_PIX) {
        float4 x = v[i];
        if (!((msk >> i) & 1u)) {
          x = make_float4(0.f, 0.f, 0.f, 0.f);  // padding is exactly zero: it pads the ACTIVATED tensor
        } else if (pro) {
          x.x = silu_fast(fmaf(ca.x, x.x, cb.x));
          x.y = silu_fast(fmaf(ca.y, x.y, cb.y));
          x.z = silu_fast(fmaf(ca.z, x.z, cb.z));
          x.w = silu_fast(fmaf(ca.w, x.w, cb.w));
        }
        uint2 p1, p2, p3;
        split4(x, p1, p2, p3);
        unsigned char* dst = in_tile + wr0 + i * 32 * PITCH;
        *reinterpret_cast<uint2*>(dst) = p1;
        *reinterpret_cast<uint2*>(dst + 64) = p2;
        *reinterpret_cast<uint2*>(dst + 128) = p3;
      }
    }
    __syncthreads();
    // the next chunk's tile travels during this chunk's matrix phase (last chunk: harmless re-load of itself)
    issue_chunk_loads(ch + 1 < nchunks ? ch + 1 : ch);
    __builtin_amdgcn_sched_barrier(0);

#ifdef DMH_STAMPS
    if (!(p.ablate & 2))
#endif
    for (int tap = 0; tap < NTAPS; ++tap) {
      const int kh = tap / KW, kw = tap % KW;
      const unsigned char* at = in_tile + (kh * IN_W + kw) * PITCH;
#pragma unroll
      for (int k16 = 0; k16 < 2; ++k16) {
        if (!(ABL & 1)) load_b(k16 ^ 1, step + 1);  // prefetch the next K step's weights (other buffer)
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 2) || tap == 0)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            a[mb][s] = *reinterpret_cast<const bf16x8*>(at + arow[mb] + s * 64 + k16 * 32);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            // smallest terms first
#define DMH_TERM(sa, sb)                                                                                   \
  acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][sa], __builtin_bit_cast(bf16x8, bq[k16][nb * 3 + sb]), \
                                                        acc[mb][nb], 0, 0, 0);
            if (!(ABL & 4)) {
              DMH_TERM(2, 0)
              DMH_TERM(0, 2)
              DMH_TERM(1, 1)
            }
            DMH_TERM(1, 0)
            DMH_TERM(0, 1)
            DMH_TERM(0, 0)
#undef DMH_TERM
          }
        ++step;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: accumulators -> LDS transpose -> rows (conv_args.h)
#ifdef DMH_STAMPS
  if (!(p.ablate & 4))
#endif
  {
    constexpr int EP = EpilogueRows::EP;
    float* wl = lds + wave * (32 * EP);
    EpilogueRows er(p, b, n0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      __syncthreads();
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * half) * EP + nb * 32 + l31] = acc[mb][nb][r];
      __syncthreads();
      er.template store_rows<TW>(p, wl, wm * 64 + mb * 32, oy0, ox0);
    }
    // GroupNorm partials per 8-row x 16-column stat tile: waves sharing (stat tile, channel block) reduce together
    er.template write_stats_grid<WM, WN, TH>(p, lds, ty, tx);
  }
}

// ------------------------------------------------------------------------------ weight packing
// bf16 element index: ((((((nt * nchunks + ch) * NTAPS + tap) * 2 + k16) * 2 + nb) * 3 + piece) * 64 + lane) * 8 + j
//   -> piece of w[o = nt*64 + nb*32 + (lane & 31)][c = chunk channel k16*16 + (lane >> 5)*8 + j][tap]
__global__ void pack_bf16x3_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout,
                                          int C0, int C1, int NTAPS, int nch0, int nch1, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t r = idx;
  const int j = r % 8;
  r /= 8;
  const int lane = r % 64;
  r /= 64;
  const int piece = r % 3;
  r /= 3;
  const int nb = r % 2;
  r /= 2;
  const int k16 = r % 2;
  r /= 2;
  const int tap = r % NTAPS;
  r /= NTAPS;
  const int ch = r % (nch0 + nch1);
  const int nt = r / (nch0 + nch1);
  const int o = nt * 64 + nb * 32 + (lane & 31);
  const int k = k16 * 16 + (lane >> 5) * 8 + j;
  int c;
  bool ok;
  if (ch < nch0) {
    c = ch * KC + k;
    ok = c < C0;
  } else {
    c = (ch - nch0) * KC + k;
    ok = c < C1;
    c += C0;
  }
  float val = 0.f;
  if (ok && o < Cout) val = w[((size_t)o * (C0 + C1) + c) * NTAPS + tap];
  const float a1 = __uint_as_float(__float_as_uint(val) & 0xffff0000u);
  const float r1 = val - a1;
  const float a2 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
  const float r2 = r1 - a2;
  const float pc = piece == 0 ? val : (piece == 1 ? r1 : r2);
  wp[idx] = (unsigned short)(__float_as_uint(pc) >> 16);
}

int64_t dmh_bf16x3_pack_floats(int Cout, int C0, int C1, int KH, int KW) {
  // three bf16 per weight = 1.5 floats
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * KH * KW * 64 * KC * 3 / 2;
}

int dmh_bf16x3_pack(const float* w, float* wpack, int Cout, int C0, int C1, int KH, int KW, hipStream_t st) {
  const int nch0 = cdiv(C0, KC), nch1 = cdiv(C1, KC);
  const int64_t total = dmh_bf16x3_pack_floats(Cout, C0, C1, KH, KW) * 2;
  hipLaunchKernelGGL(pack_bf16x3_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w,
                     reinterpret_cast<unsigned short*>(wpack), Cout, C0, C1, KH * KW, nch0, nch1, total);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight");
  return DMH_OK;
}

template <int KH, int KW, int S, int UPS, int TH, int TW, int WM, int WN, int ABL = 0>
static int launch_split(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = SplitCfg<KH, KW, S, UPS, TH, TW, WM, WN>;
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, TW);
#ifdef DMH_STAMPS
  {
    const char* e = getenv("DMH_WINO_ABLATE");
    a.ablate = e ? atoi(e) : 0;
  }
#endif
  auto kern = conv_bf16x3_kernel<KH, KW, S, UPS, TH, TW, WM, WN, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Cfg::LDS_BYTES);
    DMH_REQUIRE(e == hipSuccess, "dmh_conv2d: cannot raise the LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64 * WN));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d");
  return DMH_OK;
}

// 3x3 stride 1 (optionally behind a nearest x2 upsample).  Cout a multiple of 128: 8x16 pixels x 128 channels
// per workgroup (half the staging per flop); otherwise 16x16 pixels x 64 channels.
int dmh_bf16x3_launch3(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  const bool wide = d->Cout % 128 == 0;
#ifdef DMH_STAMPS
  if (const char* e = getenv("DMH_BX_ABL")) {
    switch (atoi(e)) {
      case 1: return wide ? launch_split<3, 3, 1, 0, 8, 16, 2, 2, 1>(d, Hout, Wout, st) : launch_split<3, 3, 1, 0, 16, 16, 4, 1, 1>(d, Hout, Wout, st);
      case 2: return wide ? launch_split<3, 3, 1, 0, 8, 16, 2, 2, 2>(d, Hout, Wout, st) : launch_split<3, 3, 1, 0, 16, 16, 4, 1, 2>(d, Hout, Wout, st);
      case 3: return wide ? launch_split<3, 3, 1, 0, 8, 16, 2, 2, 3>(d, Hout, Wout, st) : launch_split<3, 3, 1, 0, 16, 16, 4, 1, 3>(d, Hout, Wout, st);
      case 4: return wide ? launch_split<3, 3, 1, 0, 8, 16, 2, 2, 4>(d, Hout, Wout, st) : launch_split<3, 3, 1, 0, 16, 16, 4, 1, 4>(d, Hout, Wout, st);
      case 7: return wide ? launch_split<3, 3, 1, 0, 8, 16, 2, 2, 7>(d, Hout, Wout, st) : launch_split<3, 3, 1, 0, 16, 16, 4, 1, 7>(d, Hout, Wout, st);
    }
  }
#endif
  if (d->upsample2) {
    return wide ? launch_split<3, 3, 1, 1, 8, 16, 2, 2>(d, Hout, Wout, st)
                : launch_split<3, 3, 1, 1, 16, 16, 4, 1>(d, Hout, Wout, st);
  }
  return wide ? launch_split<3, 3, 1, 0, 8, 16, 2, 2>(d, Hout, Wout, st)
              : launch_split<3, 3, 1, 0, 16, 16, 4, 1>(d, Hout, Wout, st);
}
