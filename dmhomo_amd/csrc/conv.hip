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
// Epilogue: accumulators -> LDS transpose -> + bias, + residual (optionally through
// SiLU(a*res+b)), 16 B-per-lane NHWC stores, and the per-tile (sum, sum^2) per output channel
// that GroupNorm needs (fixed order, no atomics).
//
// conv_igemm_kernel serves 1x1, 7x7, 4x4/s2, 2x2/s2 (and 3x3 when DMH_CONV3_VARIANT selects it); the 3x3
// stride-1 hot path defaults to the Winograd kernel of conv_wino.hip, which shares this epilogue.
//
// Replaces: F.conv2d of WeightStandardizedConv2d CFG:128, Downsample CFG:110-111 /
// DDP:110-113, Upsample CFG:106-107, to_qkv / to_out / res_conv 1x1 convs, init_conv CFG:333.
#include <stdlib.h>

#include "common.h"

#include "conv_args.h"

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
  static constexpr int EPI_BYTES = 4 * 32 * 68 * 4;  // epilogue transpose scratch: 4 waves x 32 rows x (64+4)
  static constexpr int MAIN_BYTES = (IN_FLOATS + 2 * W_FLOATS) * 4;
  static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};

// ---- shared epilogue.  C/D layout of a 32x32 block: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Each wave transposes its rows through a private LDS slab so that 16 lanes cover one pixel's 64 channels
// with float4 accesses (bias / residual / stores all 16 B per lane): conv_args.h::EpilogueRows.
template <int MB, int TH, int TW>
__device__ __forceinline__ void conv_epilogue(floatx16 (&acc)[MB][2], const ConvArgs& p, float* lds, int b, int n0,
                                              int oy0, int ox0, int tile_in_sample) {
  constexpr int EP = EpilogueRows::EP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  float* wl = lds + wave * (32 * EP);
  EpilogueRows er(p, b, n0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    __syncthreads();  // main loop (or the previous half) is done with this LDS
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * half) * EP + nb * 32 + l31] = acc[mb][nb][r];
    __syncthreads();
    er.template store_rows<TW>(p, wl, wave * (MB * 32) + mb * 32, oy0, ox0);
  }
  er.write_stats(p, lds, tile_in_sample);
}

template <int KH, int KW, int S, int UPS, int KC, int TH, int TW, int WPE>
__global__ __launch_bounds__(256, WPE) void conv_igemm_kernel(ConvArgs p) {
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
  if (t / p.tilesY >= dmh_rows_n(p.rows, p.B)) return;  // (DmhConv.rows: the workgroups of inactive rows retire)
  const int b = dmh_rows_phys(p.rows, t / p.tilesY);
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

  // input (halo) tile of one channel chunk: global -> registers (issued one phase ahead of its use).
  // Pixel offsets and the validity mask do not depend on the chunk: computed once (every vector instruction
  // in the chunk loop competes with the fp32 matrix pipe for the same ALUs).  Loads are UNCONDITIONAL from a
  // clamped, always valid address and zeroed afterwards by the mask, so they stay in flight under counted waits.
  float4 v[NLOAD];
  float4 ca, cb;
  int poff[NLOAD];
  unsigned inside = 0;  // bit i: slot i holds image data (not zero padding)
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int pix = (tid + i * 256) / C4;
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
    if (p.in_coef != nullptr && !s1) {  // wave-uniform condition; clamped channel for the padded tail
      ca = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + cc);
      cb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + cc);
    }
  };
  issue_chunk_loads(0);
  const int wr_in0 = (tid / C4) * KCP + c4 * 4;       // LDS offset of slot 0; slot i is 256 / C4 pixels further
  const int wr_w0 = (tid / C4) * KCP + (tid % C4) * 4;  // weight-tile slot 0; slot i is 256 / C4 columns further

  for (int ch = 0; ch < nchunks; ++ch) {
    const bool s1c = ch >= p.nch0;
    const bool pro = (p.in_coef != nullptr) && !s1c;
    const bool cvalid = (s1c ? ch - p.nch0 : ch) * KC + c4 * 4 < (s1c ? p.C1 : p.C0);
    const unsigned msk = cvalid ? inside : 0u;
    __syncthreads();  // every wave is done reading in_tile of the previous chunk
    // ---- registers -> (prologue SiLU(a*x+b)) -> LDS
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if ((i + 1) * 256 <= IN_PIX * C4 || (tid + i * 256) / C4 < IN_PIX) {
        float4 x = v[i];
        if (!((msk >> i) & 1u)) {
          x = make_float4(0.f, 0.f, 0.f, 0.f);  // padding stays exactly zero: it pads the ACTIVATED tensor
        } else if (pro) {
          x.x = silu_fast(fmaf(ca.x, x.x, cb.x));
          x.y = silu_fast(fmaf(ca.y, x.y, cb.y));
          x.z = silu_fast(fmaf(ca.z, x.z, cb.z));
          x.w = silu_fast(fmaf(ca.w, x.w, cb.w));
        }
        st4(in_tile + wr_in0 + i * (256 / C4) * KCP, x);
      }
    }

    for (int tap = 0; tap < NTAPS; ++tap) {
      // ---- publish the prefetched weight tile, prefetch the next one
      float* wt = w_tile + wbuf * Cfg::W_FLOATS;
#pragma unroll
      for (int i = 0; i < WL; ++i) st4(wt + wr_w0 + i * (256 / C4) * KCP, wreg[i]);
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
        // the next chunk's input tile travels during the last tap phase of this one
        // (last chunk: a harmless re-load of itself — an unconditional issue lets hipcc count vmcnt exactly)
        if (tap == NTAPS - 1) issue_chunk_loads(ch + 1 < nchunks ? ch + 1 : ch);
      }
      // keep the prefetch loads ABOVE the matrix phase (hipcc otherwise sinks them next to their
      // consumer and the global latency lands on the critical path of every phase)
      __builtin_amdgcn_sched_barrier(0);
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
      __builtin_amdgcn_sched_barrier(0);
      wbuf ^= 1;
    }
  }

  conv_epilogue<MB, TH, TW>(acc, p, lds, b, n0, oy0, ox0, tile_in_sample);
}

// ------------------------------------------------------------------------------ weight packing
// [ntile][chunk][tap][col 64][k KC]: one (chunk, tap) weight tile is a linear copy into LDS
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
template <int KH, int KW, int S, int UPS, int KC, int TH, int TW, int WPE = 2>
static int launch_conv(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = ConvCfg<KH, KW, S, UPS, KC, TH, TW>;
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, TW);
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
  hipLaunchKernelGGL((conv_igemm_kernel<KH, KW, S, UPS, KC, TH, TW, WPE>), grid, dim3(256), Cfg::LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d");
  return DMH_OK;
}

// 3x3 tiling variants (development knob DMH_CONV3_VARIANT, read once; the default is the measured best):
//   0..3  conv_igemm_kernel   (KC,TH) = (32,16) (16,16) (32,8) (16,8)
//   6     conv_wino_kernel    Winograd F(2x2,3x3), 8x16 pixels x 64 cout per workgroup
//   (7, 8: the bf16-piece direct / Winograd kernels of round 1 were measured slower than 9 on every shape and are no
//    longer part of the library — DESIGN.md 3.1 keeps their numbers, the git history their sources)
//   9     conv_f16x3_kernel   direct implicit GEMM on the fp16 matrix cores, block-scaled 2 x 3 fp16 pieces
#define DMH_CONV3_DEFAULT 9
static int conv3_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DMH_CONV3_VARIANT");
    v = e ? atoi(e) : DMH_CONV3_DEFAULT;
    if (v < 0 || v > 9 || v == 4 || v == 5 || v == 7 || v == 8) v = DMH_CONV3_DEFAULT;
  }
  return v;
}

// the fp16-piece kernel serves 3x3 and 1x1 stride-1 convolutions when variant 9 (the default) is selected
static bool use_f16x3(int KH, int stride) {
  return conv3_variant() == 9 && stride == 1 && (KH == 3 || KH == 1 || KH == 7);
}
// ... and the 4x4 / stride-2 Downsample conv as a 2x2 conv over a space-to-depth view, when the shape allows it
static bool use_f16x3_s2d(int KH, int stride, int C0, int C1) {
  return conv3_variant() == 9 && KH == 4 && stride == 2 && C1 == 0 && C0 % 32 == 0;
}

static int conv_out_dim(int in, int KH, int stride, int ups) {
  if (ups) return in * 2;
  if (stride == 1) return in;
  return KH == 4 ? (in + 2 - 4) / 2 + 1 : in / 2;
}

static int conv_th(int KH, int stride) {
  if (stride == 2) return 8;
  if (use_f16x3(KH, stride)) return 8;  // GroupNorm partials per 8 x 16 stat tile
  if (KH == 3 && (conv3_variant() == 2 || conv3_variant() == 3 || conv3_variant() >= 5)) return 8;
  return 16;
}

static int conv_kc_v(int KH, int stride) {
  if (KH == 3 && stride == 1 && conv3_variant() != 0 && conv3_variant() != 2) return 16;
  return conv_kc(KH, stride);
}

extern "C" int dmh_conv_tiles(int Hout, int Wout, int KH, int stride) {
  if (!dmh_dims_ok({Hout, Wout})) return -1;
  return cdiv(Hout, conv_th(KH, stride)) * cdiv(Wout, 16);
}

extern "C" int64_t dmh_conv_pack_floats(int Cout, int C0, int C1, int KH, int KW) {
  if (!dmh_dims_ok({Cout, C0}) || !dmh_dims_ok({C1}, 0) || !dmh_dims_ok({KH, KW}, 1, 16)) return -1;
  if (KH == 3 && conv3_variant() == 6) return dmh_wino_pack_floats(Cout, C0, C1);
  if (use_f16x3(KH, (KH == 4 || KH == 2) ? 2 : 1) || use_f16x3_s2d(KH, 2, C0, C1))
    return dmh_f16x3_pack_floats(Cout, C0, C1, KH, KW);
  const int stride = (KH == 4 || KH == 2) ? 2 : 1;
  const int KC = conv_kc_v(KH, stride);
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * KH * KW * 64 * KC;
}

extern "C" int dmh_pack_conv_weight(const float* w, float* wpack, int Cout, int C0, int C1, int KH, int KW,
                                    void* stream) {
  DMH_REQUIRE(w && wpack && Cout > 0 && C0 > 0 && C1 >= 0, "dmh_pack_conv_weight: bad arguments");
  DMH_REQUIRE(KH == KW && (KH == 1 || KH == 2 || KH == 3 || KH == 4 || KH == 7),
              "dmh_pack_conv_weight: unsupported kernel %dx%d", KH, KW);
  if (KH == 3 && conv3_variant() == 6) return dmh_wino_pack(w, wpack, Cout, C0, C1, (hipStream_t)stream);
  if (use_f16x3(KH, (KH == 4 || KH == 2) ? 2 : 1) || use_f16x3_s2d(KH, 2, C0, C1))
    return dmh_f16x3_pack(w, wpack, Cout, C0, C1, KH, KW, (hipStream_t)stream);
  const int stride = (KH == 4 || KH == 2) ? 2 : 1;
  const int KC = conv_kc_v(KH, stride);
  const int nch0 = cdiv(C0, KC), nch1 = cdiv(C1, KC);
  const int64_t total = dmh_conv_pack_floats(Cout, C0, C1, KH, KW);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     w, wpack, Cout, C0, C1, KH, KW, KC, nch0, nch1, total);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight");
  return DMH_OK;
}

// Upsample(nearest x2) + conv3x3 in its sub-pixel form (DmhConv.upsample2 == 2): -1 floats when this build / variant does
// not offer it (the caller then packs with dmh_pack_conv_weight and upsample2 = 1)
extern "C" int64_t dmh_conv_up2_pack_floats(int Cout, int C0) {
  if (!dmh_dims_ok({Cout, C0})) return -1;
  if (!use_f16x3(3, 1) || Cout % 64 != 0 || C0 % 4 != 0) return -1;
  return dmh_f16x3_up2_pack_floats(Cout, C0);
}
extern "C" int dmh_pack_conv_weight_up2(const float* w, float* wpack, int Cout, int C0, void* stream) {
  DMH_REQUIRE(w && wpack && dmh_conv_up2_pack_floats(Cout, C0) > 0, "dmh_pack_conv_weight_up2: bad arguments");
  return dmh_f16x3_up2_pack(w, wpack, Cout, C0, (hipStream_t)stream);
}

extern "C" int dmh_pack_conv_weights_multi(const DmhPackJob* jobs, int njobs, float ws_eps, void* stream) {
  DMH_REQUIRE(jobs && njobs > 0, "dmh_pack_conv_weights_multi: bad arguments");
  DMH_REQUIRE(conv3_variant() == 9, "dmh_pack_conv_weights_multi: only the fp16-piece images (DMH_CONV3_VARIANT=9)");
  for (int i = 0; i < njobs; ++i) {
    const DmhPackJob& j = jobs[i];
    DMH_REQUIRE(j.src && j.wpack && j.Cout > 0 && j.C0 > 0 && j.C1 >= 0 && (j.KH == 1 || j.KH == 3) &&
                    (j.transposed == 0 || (j.transposed == 1 && j.C1 == 0)),
                "dmh_pack_conv_weights_multi: job %d: stride-1 1x1 / 3x3 weights only (Cout %d, C0 %d, C1 %d, KH %d, transposed %d)",
                i, j.Cout, j.C0, j.C1, j.KH, j.transposed);
  }
  return dmh_f16x3_pack_multi(jobs, njobs, ws_eps, (hipStream_t)stream);
}

extern "C" int dmh_ws_standardize(const float* w, float* w_out, int Cout, int K, float eps, void* stream) {
  DMH_REQUIRE(w && w_out && Cout > 0 && K > 0, "dmh_ws_standardize: bad arguments");
  hipLaunchKernelGGL(ws_standardize_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, w_out, K, eps);
  DMH_CHECK_LAUNCH("dmh_ws_standardize");
  return DMH_OK;
}

extern "C" int dmh_conv2d(const DmhConv* d, void* stream) {
  DMH_REQUIRE(d && d->struct_size == sizeof(DmhConv),
              "dmh_conv2d: DmhConv.struct_size is %llu, this library's DmhConv has %llu bytes (header / library mismatch)",
              d ? (unsigned long long)d->struct_size : 0ull, (unsigned long long)sizeof(DmhConv));
  DMH_REQUIRE(d->src0 && d->wpack, "dmh_conv2d: null pointer");
  DMH_REQUIRE(d->B > 0 && d->Hin > 0 && d->Win > 0 && d->C0 > 0 && d->Cout > 0, "dmh_conv2d: bad shape");
  DMH_REQUIRE(d->C0 % 4 == 0 && (!d->src1 || d->C1 % 4 == 0),
              "dmh_conv2d: input channels must be a multiple of 4 (got %d, %d)", d->C0, d->C1);
  DMH_REQUIRE(!(d->in_coef && d->src1), "dmh_conv2d: the GroupNorm prologue applies to a single source");
  DMH_REQUIRE(!(d->res_coef && !d->res), "dmh_conv2d: res_coef without res");
  DMH_REQUIRE(d->Cout % 4 == 0, "dmh_conv2d: Cout must be a multiple of 4 (got %d)", d->Cout);
  DMH_REQUIRE(!d->in_bound || (d->in_coef && d->in_bound_n > 0 && d->in_bound_n <= 64),
              "dmh_conv2d: in_bound needs in_coef and 1..64 bounds per sample (got %d)", d->in_bound_n);
  hipStream_t st = (hipStream_t)stream;
  const int Hout = conv_out_dim(d->Hin, d->KH, d->stride, d->upsample2);
  const int Wout = conv_out_dim(d->Win, d->KW, d->stride, d->upsample2);
  const int key = d->KH * 100 + d->stride * 10 + d->upsample2;
  DMH_REQUIRE(d->KH == d->KW, "dmh_conv2d: non-square kernel");
  if (d->fin_w) {
    DMH_REQUIRE(d->fin_out && d->fin_n > 0 && d->fin_n <= 8, "dmh_conv2d: fin_w needs fin_out and 1 <= fin_n <= 8 (got %d)",
                d->fin_n);
    DMH_REQUIRE(d->KH == 1 && d->stride == 1 && d->Cout <= 64 && use_f16x3(1, 1),
                "dmh_conv2d: the fused final projection rides on the 1x1 fp16-piece kernel with Cout <= 64 (got %dx%d, Cout %d)",
                d->KH, d->KW, d->Cout);
  } else {
    DMH_REQUIRE(d->out, "dmh_conv2d: null pointer");
  }
  DMH_REQUIRE(!d->pix_stats || (d->KH == 1 && d->stride == 1 && d->Cout == 64 && use_f16x3(1, 1)),
              "dmh_conv2d: pix_stats rides on the 1x1 fp16-piece kernel with Cout == 64 (got %dx%d, Cout %d)", d->KH, d->KW,
              d->Cout);
  switch (key) {
    case 110:
      if (use_f16x3(1, 1)) return dmh_f16x3_launch(d, Hout, Wout, st);
      return launch_conv<1, 1, 1, 0, 32, 16, 16>(d, Hout, Wout, st);
    case 310:
      switch (conv3_variant()) {
        case 1: return launch_conv<3, 3, 1, 0, 16, 16, 16, 3>(d, Hout, Wout, st);
        case 2: return launch_conv<3, 3, 1, 0, 32, 8, 16, 3>(d, Hout, Wout, st);
        case 3: return launch_conv<3, 3, 1, 0, 16, 8, 16, 4>(d, Hout, Wout, st);
        case 6: return dmh_wino_launch(d, Hout, Wout, st);
        case 9: return dmh_f16x3_launch(d, Hout, Wout, st);
        default: return launch_conv<3, 3, 1, 0, 32, 16, 16, 2>(d, Hout, Wout, st);
      }
    case 311:
      switch (conv3_variant()) {
        case 1: return launch_conv<3, 3, 1, 1, 16, 16, 16, 3>(d, Hout, Wout, st);
        case 2: return launch_conv<3, 3, 1, 1, 32, 8, 16, 3>(d, Hout, Wout, st);
        case 3: return launch_conv<3, 3, 1, 1, 16, 8, 16, 4>(d, Hout, Wout, st);
        case 6: return dmh_wino_launch(d, Hout, Wout, st);
        case 9: return dmh_f16x3_launch(d, Hout, Wout, st);
        default: return launch_conv<3, 3, 1, 1, 32, 16, 16, 2>(d, Hout, Wout, st);
      }
    case 312:  // Upsample + 3x3 with a weight packed by dmh_pack_conv_weight_up2: four 2x2 sub-pixel convs
      DMH_REQUIRE(use_f16x3(3, 1) && !d->src1 && d->Cout % 64 == 0 && !d->stats,
                  "dmh_conv2d: the sub-pixel upsampling conv needs the fp16-piece kernels, one source, Cout %% 64 == 0, no stats");
      return dmh_f16x3_launch_up2(d, Hout, Wout, st);
    case 710:
      if (use_f16x3(7, 1)) return dmh_f16x3_launch(d, Hout, Wout, st);
      return launch_conv<7, 7, 1, 0, 16, 16, 16>(d, Hout, Wout, st);
    case 420:
      if (use_f16x3_s2d(4, 2, d->C0, d->src1 ? d->C1 : 0)) {
        DMH_REQUIRE(!d->in_coef, "dmh_conv2d: the 4x4 / stride-2 conv takes no GroupNorm prologue");
        return dmh_f16x3_launch(d, Hout, Wout, st);
      }
      return launch_conv<4, 4, 2, 0, 16, 8, 16>(d, Hout, Wout, st);
    case 220:
      DMH_REQUIRE(d->Hin % 2 == 0 && d->Win % 2 == 0, "dmh_conv2d: 2x2/s2 needs even input size");
      return launch_conv<2, 2, 2, 0, 16, 8, 16>(d, Hout, Wout, st);
    default:
      dmh_set_error("dmh_conv2d: unsupported variant k=%d stride=%d upsample=%d", d->KH, d->stride, d->upsample2);
      return DMH_EINVAL;
  }
}
