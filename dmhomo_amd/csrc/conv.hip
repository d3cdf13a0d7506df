// K1/K2 — convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
//   M = output pixels (a TH x TW tile of one sample per workgroup)
//   N = output channels (64 per workgroup)
//   K = taps x input channels, walked chunk (KC channels) by chunk, tap by tap
//
// Per input-channel chunk the (halo) input tile is staged ONCE into LDS — through the
// fused prologue SiLU(a*x+b) when the producer was a Block whose GroupNorm is still
// pending — and reused by all KH*KW taps; the weight tile of each tap is double buffered.
// The contraction runs on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain), four
// waves side by side along M, each owning MB x 2 accumulator blocks of 32x32.
// Epilogue: + bias, + residual (optionally through SiLU(a*res+b)), NHWC store, and the
// per-tile (sum, sum^2) per output channel that GroupNorm needs (fixed order, no atomics).
//
// Replaces: F.conv2d of WeightStandardizedConv2d CFG:128, Downsample CFG:110-111 /
// DDP:110-113, Upsample CFG:106-107, to_qkv / to_out / res_conv 1x1 convs, init_conv CFG:333.
#include "common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* src0;
  const float* src1;
  const float* wpack;
  const float* bias;
  const float* in_coef;
  const float* res;
  const float* res_coef;
  float* out;
  float* stats;
  int B, Hin, Win, C0, C1, Cout, Hout, Wout;
  int nch0, nch1, tilesX, tilesY;
};

template <int KH, int KW, int S, int UPS, int KC, int TH, int TW>
struct ConvCfg {
  static constexpr int BM = TH * TW;
  static constexpr int MB = BM / 128;  // 32-row MFMA blocks per wave (4 waves along M)
  static constexpr int IN_H = (TH - 1) * S + KH;
  static constexpr int IN_W = (TW - 1) * S + KW;
  static constexpr int IN_PIX = IN_H * IN_W;
  static constexpr int KCP = KC + 4;  // LDS row pitch: +16 B keeps ds_read_b128 conflict-free
  static constexpr int C4 = KC / 4;
  static constexpr int NLOAD = (IN_PIX * C4 + 255) / 256;
  static constexpr int WL = 64 * C4 / 256;
  static constexpr int PAD = (S == 1) ? (KH / 2) : (KH == 4 ? 1 : 0);
  static constexpr int IN_FLOATS = IN_PIX * KCP;
  static constexpr int W_FLOATS = 64 * KCP;
  static constexpr int LDS_BYTES = (IN_FLOATS + 2 * W_FLOATS) * 4;
};

template <int KH, int KW, int S, int UPS, int KC, int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs p) {
  using Cfg = ConvCfg<KH, KW, S, UPS, KC, TH, TW>;
  constexpr int MB = Cfg::MB, IN_W = Cfg::IN_W, IN_PIX = Cfg::IN_PIX, KCP = Cfg::KCP, C4 = Cfg::C4;
  constexpr int NLOAD = Cfg::NLOAD, WL = Cfg::WL, NTAPS = KH * KW;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* in_tile = lds;
  float* w_tile = lds + Cfg::IN_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  int t = blockIdx.x;
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY;
  const int b = t / p.tilesY;
  const int nt = blockIdx.y;
  const int n0 = nt * 64;
  const int tile_in_sample = ty * p.tilesX + tx;

  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - Cfg::PAD, ix0 = ox0 * S - Cfg::PAD;
  const int Hlim = UPS ? p.Hin * 2 : p.Hin;
  const int Wlim = UPS ? p.Win * 2 : p.Win;

  // LDS float offset of this lane's A rows (tap (0,0)), one per 32-row block
  int arow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int r = wave * (MB * 32) + mb * 32 + l31;
    const int py = r / TW, px = r % TW;
    arow[mb] = ((py * S) * IN_W + px * S) * KCP + half * 4;
  }
  const int brow = l31 * KCP + half * 4;

  floatx16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const int nchunks = p.nch0 + p.nch1;
  const float* wbase = p.wpack + (size_t)nt * nchunks * NTAPS * 64 * KC;

  // weight tile prefetch registers
  float4 wreg[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) wreg[i] = ld4(wbase + (size_t)(tid + i * 256) * 4);

  const int c4 = tid % C4;  // 256 % C4 == 0: a thread keeps its channel quad across its pixels
  int wbuf = 0;

  for (int ch = 0; ch < nchunks; ++ch) {
    const bool s1 = ch >= p.nch0;
    const float* src = s1 ? p.src1 : p.src0;
    const int Csrc = s1 ? p.C1 : p.C0;
    const int c = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4;
    const bool cvalid = c < Csrc;

    // ---- stage the input (halo) tile of this channel chunk: global -> regs -> (prologue) -> LDS
    float4 v[NLOAD];
    unsigned inside = 0;  // bit i: slot i holds image data (not zero padding)
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int f = tid + i * 256;
      const int pix = f / C4;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pix < IN_PIX && cvalid) {
        const int hy = pix / IN_W, hx = pix % IN_W;
        const int yy = iy0 + hy, xx = ix0 + hx;
        if (yy >= 0 && yy < Hlim && xx >= 0 && xx < Wlim) {
          const int sy = UPS ? (yy >> 1) : yy, sx = UPS ? (xx >> 1) : xx;
          v[i] = ld4(src + ((size_t)(b * p.Hin + sy) * p.Win + sx) * Csrc + c);
          inside |= 1u << i;
        }
      }
    }
    float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool pro = (p.in_coef != nullptr) && !s1;
    if (pro && cvalid) {
      ca = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + c);
      cb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + c);
    }
    __syncthreads();  // every wave is done reading in_tile of the previous chunk
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int f = tid + i * 256;
      const int pix = f / C4;
      if (pix < IN_PIX) {
        float4 x = v[i];
        if (pro && ((inside >> i) & 1u)) {  // padding stays exactly zero: it pads the ACTIVATED tensor
          x.x = silu_f(fmaf(ca.x, x.x, cb.x));
          x.y = silu_f(fmaf(ca.y, x.y, cb.y));
          x.z = silu_f(fmaf(ca.z, x.z, cb.z));
          x.w = silu_f(fmaf(ca.w, x.w, cb.w));
        }
        st4(in_tile + pix * KCP + c4 * 4, x);
      }
    }

    for (int tap = 0; tap < NTAPS; ++tap) {
      // ---- publish the prefetched weight tile, prefetch the next one
      float* wt = w_tile + wbuf * Cfg::W_FLOATS;
#pragma unroll
      for (int i = 0; i < WL; ++i) {
        const int f = tid + i * 256;
        st4(wt + (f / C4) * KCP + (f % C4) * 4, wreg[i]);
      }
      __syncthreads();
      {
        int nch = ch, ntap = tap + 1;
        if (ntap == NTAPS) {
          ntap = 0;
          nch = ch + 1;
        }
        if (nch < nchunks) {
          const float* wsrc = wbase + ((size_t)nch * NTAPS + ntap) * 64 * KC;
#pragma unroll
          for (int i = 0; i < WL; ++i) wreg[i] = ld4(wsrc + (size_t)(tid + i * 256) * 4);
        }
      }
      const int kh = tap / KW, kw = tap % KW;
      const float* at = in_tile + (kh * IN_W + kw) * KCP;
      const float* bt = wt + brow;
#pragma unroll
      for (int k8 = 0; k8 < KC / 8; ++k8) {
        float4 a[MB], bq[2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = ld4(at + arow[mb] + k8 * 8);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bq[nb] = ld4(bt + nb * 32 * KCP + k8 * 8);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].x, bq[nb].x, acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].y, bq[nb].y, acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].z, bq[nb].z, acc[mb][nb], 0, 0, 0);
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].w, bq[nb].w, acc[mb][nb], 0, 0, 0);
          }
      }
      wbuf ^= 1;
    }
  }

  // ---- epilogue.  C/D layout of the 32x32 block: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int chn = n0 + nb * 32 + l31;
    const bool cok = chn < p.Cout;
    const float bias = (cok && p.bias) ? p.bias[chn] : 0.f;
    float ra = 1.f, rb = 0.f;
    if (cok && p.res_coef) {
      ra = p.res_coef[(size_t)(b * 2 + 0) * p.Cout + chn];
      rb = p.res_coef[(size_t)(b * 2 + 1) * p.Cout + chn];
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * (MB * 32) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oy = oy0 + row / TW, ox = ox0 + row % TW;
        if (cok && oy < p.Hout && ox < p.Wout) {
          const size_t o = ((size_t)(b * p.Hout + oy) * p.Wout + ox) * p.Cout + chn;
          float val = acc[mb][nb][r] + bias;
          if (p.res) {
            const float rv = p.res[o];
            val += p.res_coef ? silu_f(fmaf(ra, rv, rb)) : rv;
          }
          p.out[o] = val;
          s1[nb] += val;
          s2[nb] = fmaf(val, val, s2[nb]);
        }
      }
    }
  }
  if (p.stats) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      s1[nb] += __shfl_xor(s1[nb], 32);
      s2[nb] += __shfl_xor(s2[nb], 32);
    }
    __syncthreads();  // all waves are done with in_tile: reuse it as the cross-wave scratch
    float* red = lds;
    if (half == 0) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        red[(wave * 64 + nb * 32 + l31) * 2 + 0] = s1[nb];
        red[(wave * 64 + nb * 32 + l31) * 2 + 1] = s2[nb];
      }
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Cout) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a0 += red[(w * 64 + tid) * 2 + 0];
        a1 += red[(w * 64 + tid) * 2 + 1];
      }
      float* st = p.stats + ((size_t)(b * p.tilesX * p.tilesY + tile_in_sample) * p.Cout + n0 + tid) * 2;
      st[0] = a0;
      st[1] = a1;
    }
  }
}

// ------------------------------------------------------------------------------ weight packing
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C0,
                                        int C1, int KH, int KW, int KC, int nch0, int nch1, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t r = idx;
  const int k = r % KC;
  r /= KC;
  const int j = r % 64;
  r /= 64;
  const int tap = r % (KH * KW);
  r /= (KH * KW);
  const int ch = r % (nch0 + nch1);
  const int nt = r / (nch0 + nch1);
  const int o = nt * 64 + j;
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
  if (ok && o < Cout) val = w[((size_t)o * (C0 + C1) + c) * (KH * KW) + tap];
  wp[idx] = val;
}

// N1: per-output-channel weight standardisation (two-pass, one workgroup per channel)
__global__ __launch_bounds__(256) void ws_standardize_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                             int K, float eps) {
  __shared__ float red[8];
  const int o = blockIdx.x;
  const float* wr = w + (size_t)o * K;
  float s = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) s += wr[i];
  for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)K;
  float q = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) {
    const float d = wr[i] - mean;
    q = fmaf(d, d, q);
  }
  for (int off = 32; off; off >>= 1) q += __shfl_xor(q, off);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = q;
  __syncthreads();
  const float var = (red[4] + red[5] + red[6] + red[7]) / (float)K;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int i = threadIdx.x; i < K; i += 256) out[(size_t)o * K + i] = (wr[i] - mean) * rstd;
}

// ------------------------------------------------------------------------------ host side
template <int KH, int KW, int S, int UPS, int KC, int TH, int TW>
static int launch_conv(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = ConvCfg<KH, KW, S, UPS, KC, TH, TW>;
  ConvArgs a;
  a.src0 = d->src0;
  a.src1 = d->src1;
  a.wpack = d->wpack;
  a.bias = d->bias;
  a.in_coef = d->in_coef;
  a.res = d->res;
  a.res_coef = d->res_coef;
  a.out = d->out;
  a.stats = d->stats;
  a.B = d->B;
  a.Hin = d->Hin;
  a.Win = d->Win;
  a.C0 = d->C0;
  a.C1 = d->src1 ? d->C1 : 0;
  a.Cout = d->Cout;
  a.Hout = Hout;
  a.Wout = Wout;
  a.nch0 = cdiv(a.C0, KC);
  a.nch1 = cdiv(a.C1, KC);
  a.tilesX = cdiv(Wout, TW);
  a.tilesY = cdiv(Hout, TH);
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
  hipLaunchKernelGGL((conv_igemm_kernel<KH, KW, S, UPS, KC, TH, TW>), grid, dim3(256), Cfg::LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d");
  return DMH_OK;
}

static int conv_out_dim(int in, int KH, int stride, int ups) {
  if (ups) return in * 2;
  if (stride == 1) return in;
  return KH == 4 ? (in + 2 - 4) / 2 + 1 : in / 2;
}

extern "C" int dmh_conv_tiles(int Hout, int Wout, int KH, int stride) {
  const int TH = (stride == 2) ? 8 : 16;
  (void)KH;
  return cdiv(Hout, TH) * cdiv(Wout, 16);
}

extern "C" int64_t dmh_conv_pack_floats(int Cout, int C0, int C1, int KH, int KW) {
  const int stride = (KH == 4 || KH == 2) ? 2 : 1;
  const int KC = conv_kc(KH, stride);
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * KH * KW * 64 * KC;
}

extern "C" int dmh_pack_conv_weight(const float* w, float* wpack, int Cout, int C0, int C1, int KH, int KW,
                                    void* stream) {
  DMH_REQUIRE(w && wpack && Cout > 0 && C0 > 0 && C1 >= 0, "dmh_pack_conv_weight: bad arguments");
  DMH_REQUIRE(KH == KW && (KH == 1 || KH == 2 || KH == 3 || KH == 4 || KH == 7),
              "dmh_pack_conv_weight: unsupported kernel %dx%d", KH, KW);
  const int stride = (KH == 4 || KH == 2) ? 2 : 1;
  const int KC = conv_kc(KH, stride);
  const int nch0 = cdiv(C0, KC), nch1 = cdiv(C1, KC);
  const int64_t total = dmh_conv_pack_floats(Cout, C0, C1, KH, KW);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     w, wpack, Cout, C0, C1, KH, KW, KC, nch0, nch1, total);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight");
  return DMH_OK;
}

extern "C" int dmh_ws_standardize(const float* w, float* w_out, int Cout, int K, float eps, void* stream) {
  DMH_REQUIRE(w && w_out && Cout > 0 && K > 0, "dmh_ws_standardize: bad arguments");
  hipLaunchKernelGGL(ws_standardize_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, w_out, K, eps);
  DMH_CHECK_LAUNCH("dmh_ws_standardize");
  return DMH_OK;
}

extern "C" int dmh_conv2d(const DmhConv* d, void* stream) {
  DMH_REQUIRE(d && d->src0 && d->wpack && d->out, "dmh_conv2d: null pointer");
  DMH_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->C0 > 0 && d->Cout > 0, "dmh_conv2d: bad shape");
  DMH_REQUIRE(d->C0 % 4 == 0 && (!d->src1 || d->C1 % 4 == 0),
              "dmh_conv2d: input channels must be a multiple of 4 (got %d, %d)", d->C0, d->C1);
  DMH_REQUIRE(!(d->in_coef && d->src1), "dmh_conv2d: the GroupNorm prologue applies to a single source");
  DMH_REQUIRE(!(d->res_coef && !d->res), "dmh_conv2d: res_coef without res");
  hipStream_t st = (hipStream_t)stream;
  const int Hout = conv_out_dim(d->Hin, d->KH, d->stride, d->upsample2);
  const int Wout = conv_out_dim(d->Win, d->KW, d->stride, d->upsample2);
  const int key = d->KH * 100 + d->stride * 10 + d->upsample2;
  DMH_REQUIRE(d->KH == d->KW, "dmh_conv2d: non-square kernel");
  switch (key) {
    case 110: return launch_conv<1, 1, 1, 0, 32, 16, 16>(d, Hout, Wout, st);
    case 310: return launch_conv<3, 3, 1, 0, 32, 16, 16>(d, Hout, Wout, st);
    case 311: return launch_conv<3, 3, 1, 1, 32, 16, 16>(d, Hout, Wout, st);
    case 710: return launch_conv<7, 7, 1, 0, 16, 16, 16>(d, Hout, Wout, st);
    case 420: return launch_conv<4, 4, 2, 0, 16, 8, 16>(d, Hout, Wout, st);
    case 220:
      DMH_REQUIRE(d->Hin % 2 == 0 && d->Win % 2 == 0, "dmh_conv2d: 2x2/s2 needs even input size");
      return launch_conv<2, 2, 2, 0, 16, 8, 16>(d, Hout, Wout, st);
    default:
      dmh_set_error("dmh_conv2d: unsupported variant k=%d stride=%d upsample=%d", d->KH, d->stride, d->upsample2);
      return DMH_EINVAL;
  }
}
