// K1 / K2 on the fp16 matrix cores — convolution as an implicit GEMM whose fp32 operands are carried as fp16
// pieces under block scaling ("f16x3": three MFMAs per fp32 product block, fp32 accumulation).
//
// Why: on gfx950 the fp32 MFMA executes on the vector ALUs at the fp32 VALU rate (157 TFLOP/s, no overlap with
// any other vector work: tools/micro/coexec.hip).  The 16-bit MFMAs run on the matrix cores proper at 16x that
// rate, so three of them per product block are ~5x cheaper than one fp32 MFMA.
//
// Arithmetic (all scalings are powers of two, hence exact):
//   activations  xs = x * s            s per (workgroup, running over chunks): the largest |x| staged so far sits in
//                                       [2^14, 2^15); when a later chunk raises the maximum the accumulators are rescaled
//                h1 = fp16(xs)                       11 significant bits
//                h2 = fp16(xs - h1)                  the next 11 bits; the subtraction is exact.  Stored unscaled (round 2;
//                                                    round 1 stored (xs - h1) * 2^11 and multiplied it with g1 * 2^-11,
//                                                    derived with 288 packed fp16 multiplies per wave and tile).  The
//                                                    residual of an element below 2^-3 (= block maximum * 2^-18) is an
//                                                    fp16 subnormal: its absolute error is then <= 2^-25 = block
//                                                    maximum * 2^-40, far below the fp32 accumulation rounding (2^-24 of
//                                                    the sum), so nothing measurable is lost (dynamic-range cases of
//                                                    tests/test_gpu_kernels.py: 3.6e-7..8.8e-7 before and after).  This
//                                                    is how the weights have always been split
//   weights      ws = w * 2^k          k per output channel: max |w| of the channel sits in [2^14, 2^15)   (pack time)
//                g1 = fp16(ws),  g2 = fp16(ws - g1)  (streamed)
//   product      x*w*s*2^k = h1*g1 + h1*g2 + h2*g1 + (h2*g2 + rounding of the pieces) : the dropped part is
//                <= 2^-22 relative, every fp16*fp16 product is exact in fp32, the MFMA accumulates in fp32
//   output       y = acc / s * 2^-k + bias
// 22 significant bits per operand for everything within 2^18 of its block maximum, fp32 exponent range, no overflow
// by construction.  Against an fp64 convolution the error is that of the fp32-MFMA kernels (fp32 accumulation
// rounding dominates both): tests/test_gpu_kernels.py.
//
//   M = output pixels: each wave owns 64 (four 16-row blocks = four tile rows);  N = 64 output channels per wave;
//       v_mfma_f32_16x16x32_f16 (K = the 32 channels of a chunk): at equal cycles per FLOP this shape draws less
//       power than 32x32x16, and the kernel is power-limited
//   workgroup = 4 waves as WM x WN:  4x1 -> 16x16 pixels x 64 cout,   2x2 -> 8x16 pixels x 128 cout
//   K = taps x input channels in chunks of 32 channels: per chunk the (halo) input tile is staged ONCE into LDS —
//       through the fused prologue SiLU(a*x+b) when a GroupNorm is pending — as two fp16 planes and reused by
//       all taps; the weight planes stream from L2 in fragment order, one coalesced 1 KB load per wave each,
//       prefetched one K step ahead.
//   MFMA order (round 3): the three terms of an accumulator are issued back to back as one asm statement — the matrix pipe
//       forwards the accumulator between dependent MFMAs, so it crosses the register file once per three products; under the
//       power cap that is worth +3.7 % images/s (tools/micro/mfma_chain_power.hip).  hipcc does not know those statements
//       are MFMAs: a wait-state guard pinned to every accumulator closes each matrix phase (tools/hazard_scan.py checks the
//       generated code for anything of hipcc's own that names such a register too soon).
// Epilogue: shared with conv.hip (LDS transpose, + bias, + residual, float4 NHWC stores, GroupNorm partials; for the 1x1
//       16x16 layout optionally the UNet's final projection and the LayerNorm statistics of the output: DmhConv.fin_*, pix_stats).
//
// UPS == 2 is the Downsample conv (4x4, stride 2, pad 1, CFG:110-111) as an exact 2x2 / stride-1 convolution over the
// space-to-depth view of its input shifted by one pixel: X[cy][cx][(py, px, c)] = x[2cy-1+py][2cx-1+px][c], so
// y[oy][ox] = sum_{dy,dx in {0,1}} W[2dy+py][2dx+px] X[oy+dy][ox+dx] — the view is only index math in the halo gather
// (a 'channel chunk' is then 32 channels of one of the four pixel parities), the weights are re-indexed at pack time.
//
// UPS == 4 is the 7x7 init conv (CFG:333) with at most 16 input channels: TWO taps share one K = 32 slice (K slots 0..15 =
// the 16 channels of tap 2t, 16..31 = those of tap 2t + 1: a lane's 8-channel group picks its tap), 25 tap pairs instead of
// 49 taps whose upper K half would be zero padding.
//
// Replaces: the 3x3 convolutions of Block / ResnetBlock CFG:128-170, the Upsample conv CFG:106-107, the Downsample
// conv CFG:110-111 and the 1x1 convolutions (to_qkv / to_out / res_conv) of CFG:176-245.
#include <stdlib.h>

#include "common.h"

#include "conv_args.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32;                // input channels per chunk
constexpr int PITCH = 160;            // LDS bytes per staged pixel: 2 planes x 32 fp16 + 32 (see F16Cfg)
constexpr int STEP_U4 = 4 * 64;       // uint4 per (chunk, tap, cout half): 2 column blocks x 2 planes (g1, g2) x 64 lanes

template <int KH, int KW, int S, int UPS, int TH, int TW, int WM, int WN>
struct F16Cfg {
  static_assert(WM * WN == 4, "four waves per workgroup");
  // UPS == 3: the sub-pixel form of Upsample(nearest x2) + conv3x3.  Output pixel (2y+dy, 2x+dx) only sees the 2x2
  // low-resolution pixels (y+dy-1 .. y+dy, x+dx-1 .. x+dx), through weights that are sums of the 3x3 taps falling on the
  // same source pixel: four 2x2 convs (one per parity) instead of one 3x3 at the high resolution, 16 instead of 36
  // multiply-adds per low-resolution pixel.  The tile is 4 x 16 LOW-resolution pixels; wave w owns parity (w >> 1, w & 1)
  // of all 64 of them, with its own weight stream (the packed weight holds 4*Cout virtual output channels).
  static constexpr bool SUB = UPS == 3;
  static_assert(SUB || TH * TW == WM * 64, "64 pixels per wave");
  static_assert(!SUB || (KH == 2 && KW == 2 && S == 1 && TH == 4 && TW == 16 && WM == 4 && WN == 1), "sub-pixel geometry");
  static constexpr int IN_H = SUB ? TH + 2 : (TH - 1) * S + KH;
  static constexpr int IN_W = SUB ? TW + 2 : (TW - 1) * S + KW;
  static constexpr int IN_PIX = IN_H * IN_W;
  // channel quads staged per pixel: 8 (one 32-channel chunk); UPS == 4 reads only channels 0..15 of a staged pixel (its
  // K slots 16..31 are another pixel's), so it stages 4 — half the loads, splits and LDS writes, 32 registers less
  static constexpr int QS = UPS == 4 ? 2 : 3, NQ = 1 << QS;
  static constexpr int NLOAD = (IN_PIX * NQ + 255) / 256;
  static constexpr int PAD = UPS == 2 ? 0 : ((S == 1) ? (KH / 2) : (KH == 4 ? 1 : 0));  // UPS == 2: cell (oy, ox) is tap (0,0)
  // An A fragment of v_mfma_f32_16x16x32_f16 is 16 pixels of one tile row x 4 K-groups of 16 B.  ds_read_b128 serves
  // lanes in groups of 16 that pair 8 pixels of one K-group with the complementary 8 pixels of the next K-group
  // (MI355X_MICROARCH.md, LDS): with a pixel pitch of 10 x 16 B those 16 lanes fall on 16 distinct 16 B bank groups.
  static constexpr int ROWP = IN_W * PITCH;
  static constexpr int IN_BYTES = IN_H * ROWP;
  static constexpr int EPI_BYTES = 4 * 32 * EpilogueRows::EP * 4;
  static constexpr int TILE_BYTES = IN_BYTES > EPI_BYTES ? IN_BYTES : EPI_BYTES;
  static constexpr int LDS_BYTES = TILE_BYTES + 16;  // + the two block-maximum slots
};

__device__ __forceinline__ unsigned absbits(float x) { return __float_as_uint(x) & 0x7fffffffu; }

// floats of the fragment part of a plain (stride-1 1x1 / 3x3) image with T taps: f16x3_frag_floats for device code
__device__ __forceinline__ int64_t f16x3_frag_floats_dev(int Cout, int C0, int C1, int T) {
  return (int64_t)((Cout + 63) / 64) * ((C0 + KC - 1) / KC + (C1 + KC - 1) / KC) * T * 2 * STEP_U4 * 4;
}

}  // namespace

template <int KH, int KW, int S, int UPS, int TH, int TW, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(ConvArgs p) {
  using Cfg = F16Cfg<KH, KW, S, UPS, TH, TW, WM, WN>;
  constexpr bool PK = UPS == 4;  // two taps per K slice (see the header)
  constexpr int IN_W = Cfg::IN_W, IN_PIX = Cfg::IN_PIX, NLOAD = Cfg::NLOAD, NTAPS = PK ? (KH * KW + 1) / 2 : KH * KW,
                ROWP = Cfg::ROWP;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned char* in_tile = reinterpret_cast<unsigned char*>(lds);
  unsigned* mxslot = reinterpret_cast<unsigned*>(in_tile + Cfg::TILE_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int kg = lane >> 4;   // K group (8 channels) of this lane's A / B fragment slice; row group of its C slice
  const int l15 = lane & 15;  // fragment row (pixel) / column (output channel)
  const int wm = wave / WN, wn = wave % WN;

  // Workgroups are dealt to the 8 XCDs round-robin by linear id, and each XCD has its own L2: re-deal the ids so that
  // XCD k walks the k-th contiguous run of (cout tile, sample, tile row, tile column) — vertically adjacent tiles, whose
  // halos overlap, then run at the same time on the same L2.
  int t = blockIdx.x, by = blockIdx.y, nb = gridDim.x;
  if (p.rows) {
    // a row subset (DmhConv.rows): the first (tiles per sample) x n x gridDim.y workgroups of the captured B-row grid do the
    // work of the n active rows — re-dealt over the XCDs below as a grid of that size, so every XCD keeps an equal share
    // whatever n is — and the rest retire here
    nb = p.tilesX * p.tilesY * p.rows[0];
    const int lin = t + by * (int)gridDim.x;
    if (lin >= nb * (int)gridDim.y) return;
    t = lin % nb;
    by = lin / nb;
  }
  if (p.xcd) {
    const int total = nb * gridDim.y, lin = t + by * nb;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, local = lin >> 3;
    const int nl = xcd * q + min(xcd, r) + local;
    if (p.xcd == 2) {   // cout tile innermost: the output-channel tiles of one pixel tile run side by side on ONE XCD, so its
                        // input tile is fetched into one L2 once (the <= 32^2 levels: launch_f16x3)
      by = nl % (int)gridDim.y;
      t = nl / (int)gridDim.y;
    } else {
      t = nl % nb;
      by = nl / nb;
    }
  }
  const int tx = t % p.tilesX;
  t /= p.tilesX;
  const int ty = t % p.tilesY;
  const int b = dmh_rows_phys(p.rows, t / p.tilesY);
  const int nt = UPS == 3 ? wm * (int)gridDim.y + by : by * WN + wn;  // UPS == 3: virtual channels [parity][cout tile]
  const int n0 = UPS == 3 ? by * 64 : nt * 64;                          // output channels of this wave

  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - Cfg::PAD, ix0 = ox0 * S - Cfg::PAD;
  const int Hlim = UPS == 1 ? p.Hin * 2 : (UPS == 2 ? p.Hin / 2 + 1 : p.Hin);  // UPS == 2: coarse cells 0 .. Hin/2
  const int Wlim = UPS == 1 ? p.Win * 2 : (UPS == 2 ? p.Win / 2 + 1 : p.Win);

  if (tid < 2) mxslot[tid] = 0u;
#ifdef DMH_STAMPS
  // diagnostic build only (make stamps; tools/f16_stamps.py): per-wave cycle totals of each phase over this tile's stats slot
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev, t_now;
#define STAMP(i)                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                             \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory"); \
  __builtin_amdgcn_sched_barrier(0);                                             \
  tk[i] += t_now - t_prev;                                                       \
  t_prev = t_now;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  const unsigned long long t_begin = t_prev;
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
#else
#define STAMP(i)
#endif

  // LDS byte offset of this lane's A rows at tap (0,0), plane 0: four 16-pixel row blocks per wave
  static_assert(TW == 16, "a 16-row MFMA block is one tile row");
  int arow[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
    arow[mb] = UPS == 3 ? (mb + (wm >> 1)) * ROWP + (l15 + (wm & 1)) * PITCH + kg * 16
                        : ((wm * 4 + mb) * S) * ROWP + (l15 * S) * PITCH + (PK ? (kg & 1) : kg) * 16;

  // accumulators: 4 pixel blocks x 4 channel blocks of 16x16.  The MFMA takes the WEIGHT fragment as its first operand, so
  // D = W * X^T: row 4 * (lane >> 4) + r = output channel, column lane & 15 = pixel — a lane holds 4 consecutive channels
  // of one pixel and the epilogue moves them as one 16 B piece
  float4v acc[4][4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = float4v{0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.nch0 + p.nch1;
  const int nsteps = nchunks * NTAPS * 2;
  // (the wave's weight stream starts at a wave-uniform address: kept in scalar registers, so a step's fragment loads are
  //  scalar base + lane offset and the walk through the steps costs no vector instructions)
  const uint4* wbase = reinterpret_cast<const uint4*>(p.wpack) + (size_t)__builtin_amdgcn_readfirstlane(nt) * nsteps * STEP_U4;

  // B fragments of one step = (tap, half of the 64 output channels): [column block][plane g1, g2]; NB buffers rotate
  // over the steps, so the loads run NB - 1 steps (two for 3x3) ahead of their use.
  // Stride-1 3x3 convolutions walk their taps COLUMN-major (kx outer, ky inner): the A fragment of tap (ky, kx) for row
  // block mb is tile row mb + ky shifted by kx columns, so one kx needs the 6 tile rows of the wave once (12 ds_read_b128)
  // for its three ky — 36 LDS reads per chunk instead of 72.  The packed weight keeps its [tap = ky * 3 + kx] order; only
  // the order in which the steps are fetched and multiplied changes.
  constexpr bool ROWREUSE = KH == 3 && KW == 3 && S == 1 && UPS != 3 && UPS != 2;
  constexpr int NSTEP = NTAPS * 2;                  // steps per chunk
  // NSTEP % NB == 0: a step's buffer index is static.  Three buffers (weights two steps ahead) for the 8 x 16 x 128 layout; the
  // 16 x 16 x 64 one has registers for two (round 3: with the accumulator-major MFMA order a third buffer spilled)
  constexpr int NB = (NSTEP % 3 == 0 && WN != 1) ? 3 : 2;
  uint4 bq[NB][4];
  auto tap_of = [](int pos) constexpr { return ROWREUSE ? (pos % 3) * 3 + pos / 3 : pos; };   // pos-th tap multiplied
  auto load_b = [&](int buf, int ch_, int st_) {    // st_ may run past the chunk: the first steps of the next one
    if (st_ >= NSTEP) {
      st_ -= NSTEP;
      ch_ += 1;
    }
    if (ch_ >= nchunks) ch_ = nchunks - 1;          // (past the end: a harmless re-load)
    const uint4* src = wbase + ((size_t)(ch_ * NTAPS + tap_of(st_ >> 1)) * 2 + (st_ & 1)) * STEP_U4;
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[buf][i] = src[i * 64 + lane];
  };
#pragma unroll
  for (int i = 0; i < NB - 1; ++i) load_b(i, 0, i);

  // input (halo) tile of one channel chunk: global -> registers, issued one chunk ahead.  Unconditional loads
  // from clamped addresses + a validity mask (see conv.hip) keep them in flight under counted waits.
  constexpr int QS = Cfg::QS, NQ = Cfg::NQ;
  const int c4 = tid & (NQ - 1);
  float4 v[NLOAD];
  // PACKPW (the 16x16-pixel layout of the plain 3x3 conv, which has no register to spare): one word per load — the LDS slot in
  // the low half, the source pixel RELATIVE to the tile's first (clamped) pixel in the high half (<= 17 * Win + 17 < 2^16: the
  // launcher checks); the tile's own offset rides on the scalar base address
  constexpr bool PACKPW = KH == 3 && KW == 3 && S == 1 && UPS == 0 && WN == 1;
  int poff[NLOAD], wroff[PACKPW ? 1 : NLOAD];
  const int y0c = min(max(iy0, 0), Hlim - 1), x0c = min(max(ix0, 0), Wlim - 1);
  const int pbase = PACKPW ? (b * p.Hin + y0c) * p.Win + x0c : 0;
  unsigned inside = 0;
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int pix = (tid + i * 256) >> QS;
    const int pixc = pix < IN_PIX ? pix : IN_PIX - 1;
    const int wro = (pixc / IN_W) * ROWP + (pixc % IN_W) * PITCH + c4 * 8;  // staging slot in the LDS tile
    if (!PACKPW) wroff[i] = wro;
    const int yy = iy0 + pixc / IN_W, xx = ix0 + pixc % IN_W;
    const bool ok = pix < IN_PIX && yy >= 0 && yy < Hlim && xx >= 0 && xx < Wlim;
    const int yc = min(max(yy, 0), Hlim - 1), xc = min(max(xx, 0), Wlim - 1);
    const int sy = UPS == 1 ? (yc >> 1) : yc, sx = UPS == 1 ? (xc >> 1) : xc;
    poff[i] = UPS == 2 ? ((yc << 16) | xc) : (b * p.Hin + sy) * p.Win + sx;  // UPS == 2: the coarse cell, resolved per parity
    if (PACKPW) poff[i] = (int)((unsigned)wro | ((unsigned)((yc - y0c) * p.Win + (xc - x0c)) << 16));
    inside |= (ok ? 1u : 0u) << i;
  }
  unsigned inside_ch = inside;  // validity mask of the chunk whose loads are in flight (UPS == 2: depends on the parity)
  // load i of chunk ch's tile (issue_prep(ch) first: it resolves the chunk's source and channel offset)
  const float* ld_src = p.src0;
  int ld_C = p.C0, ld_c = 0, ld_py = 0, ld_px = 0;
  auto issue_prep = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    ld_src = s1 ? p.src1 : p.src0;
    ld_C = s1 ? p.C1 : p.C0;
    if (UPS == 2) {
      // chunk = (pixel parity, 32 channels): source pixel (2*cy - 1 + py, 2*cx - 1 + px) of coarse cell (cy, cx)
      const int nchc = p.C0 / KC, par = ch / nchc;
      ld_py = par >> 1;
      ld_px = par & 1;
      ld_c = (ch - par * nchc) * KC + c4 * 4;
      inside_ch = 0u;
    } else {
      const int c = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4;
      ld_c = c < ld_C ? c : 0;
    }
  };
  auto issue_one = [&](int i) {
    const float* a;
    if (UPS == 2) {
      const int ry = 2 * (poff[i] >> 16) - 1 + ld_py, rx = 2 * (poff[i] & 0xffff) - 1 + ld_px;
      const bool ok = ((inside >> i) & 1u) && ry >= 0 && ry < p.Hin && rx >= 0 && rx < p.Win;
      const int yc = min(max(ry, 0), p.Hin - 1), xc = min(max(rx, 0), p.Win - 1);
      a = ld_src + ((size_t)(b * p.Hin + yc) * p.Win + xc) * ld_C + ld_c;
      inside_ch |= (ok ? 1u : 0u) << i;
    } else {
      if (PACKPW) a = ld_src + (size_t)pbase * ld_C + (int)(((unsigned)poff[i] >> 16) * (unsigned)ld_C + (unsigned)ld_c);
      else a = ld_src + (size_t)poff[i] * ld_C + ld_c;
    }
#ifdef DMH_STAMPS
    a = (p.ablate & 32) ? ld_src + ld_c : a;        // ablation: the tile loads hit one cached line (latency / HBM share)
#endif
    v[i] = ld4(a);
  };
  issue_prep(0);
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) issue_one(i);
  __syncthreads();                              // block-maximum slots are zeroed
#ifdef DMH_STAMPS
  // diagnostic build only (tools/gn_fold_cost.py, docs/EXPERIMENTS.md R5.4): what a gn_finalize folded into its CONSUMER would
  // add to the head of every workgroup — 8 KB of per-(tile, group) partials of the sample (any L2-resident 8 KB of it stands
  // in), reduced in f64 in a fixed order, (a, b) for every input channel into LDS, the barrier that publishes them
  if (p.ablate & 128) {
    const float* gs = p.src0 + (size_t)b * p.Hin * p.Win * p.C0;
    double q1 = 0.0, q2 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float2 t2 = *reinterpret_cast<const float2*>(gs + (size_t)(((tid & 31) + 32 * j) * 8 + (tid >> 5)) * 2);
      q1 += (double)t2.x;
      q2 += (double)t2.y;
    }
    for (int off = 16; off; off >>= 1) {
      q1 += __shfl_xor(q1, off);
      q2 += __shfl_xor(q2, off);
    }
    const double nn = (double)p.Hin * p.Win * (p.C0 / 8);
    const double mean = q1 / nn;
    double var = q2 / nn - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5)), meanf = (float)mean;
    float* tab = reinterpret_cast<float*>(in_tile);
    float* gtab = tab + 2 * 768;
    if ((tid & 31) == 0) {
      gtab[(tid >> 5) * 2 + 0] = meanf;
      gtab[(tid >> 5) * 2 + 1] = rstd;
    }
    __syncthreads();
    const float* cf = p.in_coef ? p.in_coef + (size_t)b * 2 * p.C0 : p.bias;
    for (int c = tid; c < p.C0 && c < 768; c += 256) {
      const int g8 = c / max(p.C0 / 8, 1);
      const float ga = cf ? cf[c % 64] : 1.f, be = cf ? cf[(c + 32) % 64] : 0.f;   // (gamma, beta, (scale, shift): three more L2 reads)
      const float aa = gtab[g8 * 2 + 1] * ga;
      tab[c * 2 + 0] = aa;
      tab[c * 2 + 1] = be - gtab[g8 * 2 + 0] * aa;
    }
    __syncthreads();
    const float probe = tab[(tid & 63) * 2] + tab[(tid & 63) * 2 + 1];
    asm volatile("" ::"v"(probe));
    __syncthreads();                            // (the staging below reuses this LDS: a real fold would own its table)
  }
#endif

  int e_run = 16;  // biased exponent of the running block maximum (clamped to [16, 254]); uniform
  // DmhConv.in_bound: the producer's GroupNorm statistics bound |a*x + b| >= |SiLU(a*x + b)| for every element of the sample,
  // so the block scale is known before the tile arrives: no maximum over the staged values, no LDS atomic, no wait for it
  // (an over-estimated scale costs range, not precision: see the header), one scale for all chunks, never a rescale
  bool stat = p.in_bound != nullptr;
  int e_fix = 16;
  if (stat) {
    float m = 0.f;
    for (int i = 0; i < p.in_bound_n; ++i) {
      const float v = p.in_bound[b * p.in_bound_n + i];
      m = (v < INFINITY) ? fmaxf(m, v) : INFINITY;   // +inf or NaN: no static scale for this sample (fmaxf would drop a NaN)
    }
    const unsigned mb = __builtin_amdgcn_readfirstlane(__float_as_uint(m));
    stat = mb < 0x7f000000u;   // +inf (or NaN): the producer declined to bound this sample -> search the tiles as usual
    // + 1: a factor of two for the rounding of the bound and of the prologue's own fma
    e_fix = min(max((int)(mb >> 23) + 1, 16), 254);
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    const bool s1c = ch >= p.nch0;
    const bool pro = (p.in_coef != nullptr) && !s1c;
    // the GroupNorm coefficients of this chunk's channels (L2-resident, 32 B per lane): fetched here rather than with
    // the halo prefetch, which would keep 8 more registers alive across the whole matrix phase
    float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pro) {
      const int cq = ch * KC + c4 * 4;
      const int ccq = cq < p.C0 ? cq : 0;
      ca = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + ccq);
      cb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + ccq);
    }
    const bool cvalid = UPS == 2 || (s1c ? ch - p.nch0 : ch) * KC + c4 * 4 < (s1c ? p.C1 : p.C0);
    const unsigned msk = cvalid ? inside_ch : 0u;
    // ---- values of this chunk (prologue applied, padding zeroed) stay in v[]; their largest magnitude -> LDS slot
    // The prologue runs on EVERY lane (padding lanes hold real values from clamped addresses) under the one uniform
    // condition, and the padding is cleared afterwards by a bit mask: with the lane test around it (round 1) each of the
    // NLOAD groups was its own exec-masked region — 11 branch pairs per chunk, and four exp -> rcp chains at a time with
    // nothing to interleave them with.
    if (pro) {
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        // the last load group is only partly populated (16 x 16 tile: 32 of its 256 lanes): the waves that hold none of it
        // skip its eight transcendentals — a scalar branch (the wave index is uniform), not an exec-masked region
        if ((i + 1) * 256 > IN_PIX * NQ && ((wave_u * 64 + i * 256) >> QS) >= IN_PIX) continue;
        v[i].x = silu_fast(fmaf(ca.x, v[i].x, cb.x));
        v[i].y = silu_fast(fmaf(ca.y, v[i].y, cb.y));
        v[i].z = silu_fast(fmaf(ca.z, v[i].z, cb.z));
        v[i].w = silu_fast(fmaf(ca.w, v[i].w, cb.w));
      }
    }
    // block maximum over everything the lanes hold, padding lanes included (their values are real ones from clamped
    // addresses, through the prologue like the rest: the scale only has to bound the tile, and an over-estimate costs
    // nothing — see the header); the padding itself becomes zero in the split below, which multiplies it by 0 instead of the
    // block scale.  The one exception: a wave that skipped the prologue of the partly filled last load group above holds RAW
    // pre-GroupNorm values there — all of them padding — and leaves that group out of the maximum under the same uniform
    // condition (raw activations can be orders of magnitude above SiLU(a*x + b) and would cost the tile its range).
    if (!stat) {
      float mxf = 0.f;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        if (pro && (i + 1) * 256 > IN_PIX * NQ && ((wave_u * 64 + i * 256) >> QS) >= IN_PIX) continue;
        mxf = fmaxf(fmaxf(mxf, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
      }
      unsigned mx = __float_as_uint(mxf);
      mx = wave_max_u32(mx);
      if (lane == 0) atomicMax(&mxslot[ch & 1], mx);
    }
    STAMP(0)  // wait for the halo loads + prologue + block maximum
    __syncthreads();  // block maximum complete; every wave is done reading the previous chunk's tile
    STAMP(1)  // barrier 1
    // ---- block scale: running maximum over the chunks, so the scale only ever shrinks (no overflow on rescale)
    const int e_old = e_run;
    int e_ch = e_fix;
    if (!stat) {
      const unsigned bmx = mxslot[ch & 1];
      if (tid == 0) mxslot[(ch + 1) & 1] = 0u;
      e_ch = min(max((int)(__builtin_amdgcn_readfirstlane(bmx) >> 23), 16), 254);
    }
    e_run = max(e_run, e_ch);
    const float sc = __uint_as_float((unsigned)(268 - e_run) << 23);  // largest |x| * sc in [2^14, 2^15)
    if (e_run != e_old && ch > 0) {
      const int fe = 127 + e_old - e_run;
      const float f = fe > 0 ? __uint_as_float((unsigned)fe << 23) : 0.f;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] *= f;
    }
    // ---- split into two fp16 planes -> LDS.  v[i] is free once it is split, and the next chunk's load i is issued right
    // there: a wave's loads return in order, so the weight fragments fetched after the tile loads wait for them too, and
    // issued at the head of the matrix phase (round 1) the HBM latency of the tile stalled the third weight step of every
    // chunk (64->64 @128^2: the matrix phase with a tile in flight took 16.7 k cycles, the one without 11.4 k)
    const bool more = ch + 1 < nchunks;
    if (more) issue_prep(ch + 1);
#ifdef DMH_STAMPS
    if (p.ablate & 1) {                             // ablation: no split / LDS write
#pragma unroll
      for (int i = 0; i < NLOAD; ++i)
        if (more) issue_one(i);
    } else
#endif
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if ((i + 1) * 256 <= IN_PIX * NQ || ((tid + i * 256) >> QS) < IN_PIX) {
        // h1 = fp16(x * sc), h2 = fp16(x * sc - h1) (exact; see the header: stored unscaled): dmh_split2, twelve
        // instructions per float4 at 43 SIMD cycles (round 2: eight v_fma_mixlo/hi_f16 at 65 — quarter-rate instructions)
        const float si = ((msk >> i) & 1u) ? sc : 0.f;   // padding is exactly zero: it pads the ACTIVATED tensor
        uint2 h1, h2;
        dmh_split2(v[i].x, v[i].y, si, h1.x, h2.x);
        dmh_split2(v[i].z, v[i].w, si, h1.y, h2.y);
        int slot = PACKPW ? poff[i] : wroff[i];
        // (volatile: hipcc otherwise hoists the eleven masks out of the chunk loop, into the registers the packing freed)
        if (PACKPW) asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(slot));
        unsigned char* dst = in_tile + slot;
        *reinterpret_cast<uint2*>(dst) = h1;
        *reinterpret_cast<uint2*>(dst + 64) = h2;
      }
      if (more) issue_one(i);
    }
    STAMP(2)  // rescale + split + LDS write
    __syncthreads();
    STAMP(3)  // barrier 2
    __builtin_amdgcn_sched_barrier(0);

    // A fragments [row block][plane] of one tap (all 32 channels of the chunk: K = 32 per MFMA)
    half8 a[4][2];
    half8 ar[ROWREUSE ? 5 : 1][2];                  // ROWREUSE: tile rows of the wave at one column shift kx; row r lives in
                                                    // slot r % 5 (ky = 0 needs rows 0-3, ky = 1 adds row 4, ky = 2 row 5)
    auto read_a = [&](int tap) {
      const unsigned char* at = in_tile + (tap / KW) * ROWP + (tap % KW) * PITCH;
      if constexpr (PK) {  // `tap` is a pair: K groups 0,1 read tap 2*tap, groups 2,3 tap 2*tap + 1 (the odd one out pairs
                           // with itself under zero weights)
        const int t0 = 2 * tap, t1 = 2 * tap + 1 < KH * KW ? 2 * tap + 1 : 2 * tap;
        const int o0 = (t0 / KW) * ROWP + (t0 % KW) * PITCH, o1 = (t1 / KW) * ROWP + (t1 % KW) * PITCH;
        at = in_tile + ((kg >> 1) ? o1 : o0);
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[mb][pl] = *reinterpret_cast<const half8*>(at + arow[mb] + pl * 64);
    };
    auto read_rows = [&](int kx, int r0, int r1) {  // tile rows wm * 4 + [r0, r1), columns l15 + kx
      const unsigned char* at = in_tile + kx * PITCH;
#pragma unroll
      for (int r = r0; r < r1; ++r)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          ar[ROWREUSE ? r % 5 : 0][pl] = *reinterpret_cast<const half8*>(at + arow[0] + r * ROWP + pl * 64);
    };
#ifdef DMH_STAMPS
    if (!(p.ablate & 2))
#endif
    {
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {          // st = 2 * (position in the tap walk) + (half of the output channels)
#ifdef DMH_STAMPS
        if (!(p.ablate & 8))                        // ablation: the weight fragments are loaded once (no B stream)
#endif
        load_b((st + NB - 1) % NB, ch, st + NB - 1);  // weights of the step NB - 1 ahead
        __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks the loads next to their use)
        const int pos = st >> 1;
#ifdef DMH_STAMPS
        if (!(p.ablate & 16) || st == 0)            // ablation: the A fragments are read once per chunk (no LDS reads)
#endif
        {
          if (ROWREUSE) {
            if ((st & 1) == 0) read_rows(pos / 3, pos % 3 == 0 ? 0 : 3 + pos % 3, 4 + pos % 3);
          } else if ((st & 1) == 0) {
            read_a(pos);
          }
        }
        auto afrag = [&](int mb, int pl) -> half8 {
          if constexpr (ROWREUSE) return ar[(mb + pos % 3) % 5][pl];
          else return a[mb][pl];
        };
#define DMH_A(mb, pl) afrag(mb, pl)
        // the three terms of ONE accumulator back to back, as one asm statement hipcc cannot pull apart: the matrix pipe
        // forwards the accumulator between dependent MFMAs (full rate: tools/micro/issue_model.hip) instead of writing it to
        // and reading it from the register file three times — fewer joules per product, and this kernel is power-limited
        // (tools/micro/mfma_chain_power.hip: the same MFMAs on the same operands, 9-10 % faster in this order).  Per
        // accumulator the order of the terms is unchanged: bitwise the same sums.
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            float4v& c = acc[mb][(st & 1) * 2 + nb];
            const half8 g1 = __builtin_bit_cast(half8, bq[st % NB][nb * 2]), g2 = __builtin_bit_cast(half8, bq[st % NB][nb * 2 + 1]);
            const half8 h1 = DMH_A(mb, 0), h2 = DMH_A(mb, 1);
            asm("v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"      // h2 * g1   (smallest terms first)
                "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"      // h1 * g2
                "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0"           // h1 * g1
                : "+v"(c) : "v"(g1), "v"(g2), "v"(h1), "v"(h2));
          }
#undef DMH_A
      }
    }
    // hipcc does not know that the asm statements above are MFMAs: nothing of its own may read an accumulator before the last
    // of them has written it (here: the rescale of the next chunk and the epilogue's slab writes, both a barrier away — the
    // guard makes the distance a fact instead of a habit).  16 wait states, twice, pinned to every accumulator.
#pragma unroll
    for (int mb = 0; mb < 4; mb += 2)
      asm volatile("s_nop 15" : "+v"(acc[mb][0]), "+v"(acc[mb][1]), "+v"(acc[mb][2]), "+v"(acc[mb][3]), "+v"(acc[mb + 1][0]),
                   "+v"(acc[mb + 1][1]), "+v"(acc[mb + 1][2]), "+v"(acc[mb + 1][3]));
    __builtin_amdgcn_sched_barrier(0);
    STAMP(4)  // matrix phase
  }

  // ---- epilogue: accumulators / block scale -> LDS transpose -> rows (conv_args.h applies 2^-k per channel)
#ifdef DMH_STAMPS
  if (!(p.ablate & 4))
#endif
  {
    constexpr int EP = EpilogueRows::EP;
    const float inv_s = __uint_as_float((unsigned)(e_run - 14) << 23);  // 1 / sc
    float* wl = lds + wave * (32 * EP);
    EpilogueRows er(p, b, n0);
    if (UPS == 3 && p.oscale) er.osc = ld4(p.oscale + nt * 64 + er.c4 * 4);  // the weight scale of the VIRTUAL channel
    er.osc.x *= inv_s;                // 1 / (block scale) rides on the per-channel 2^-k: both powers of two
    er.osc.y *= inv_s;
    er.osc.z *= inv_s;
    er.osc.w *= inv_s;
    __syncthreads();                  // every wave is done reading the input tile, which the slabs overlay
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {  // two passes of 32 pixel rows through the wave's OWN slab: a wave's LDS operations
                                      // execute in order, so nothing but the wave itself has to be waited for
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          *reinterpret_cast<float4v*>(wl + (m2 * 16 + l15) * EP + nb * 16 + kg * 4) = acc[hb * 2 + m2][nb];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (UPS == 3) er.template store_rows<TW>(p, wl, hb * 32, oy0, ox0, 2, wm >> 1, wm & 1);
      else er.template store_rows<TW, (KH == 1 && KW == 1 && WN == 1 && UPS == 0)>(p, wl, wm * 64 + hb * 32, oy0, ox0);
    }
#ifdef DMH_STAMPS
    STAMP(5)  // epilogue
    if (p.stats && lane == 0 && by == 0) {
      const int stiles = ((p.Hout + 7) / 8) * p.tilesX;
      unsigned long long* d = reinterpret_cast<unsigned long long*>(
                                  p.stats + ((size_t)(b * stiles + (ty * (TH / 8)) * p.tilesX + tx) * p.Cout) * 2) + wave * 8;
      for (int i = 0; i < 6; ++i) d[i] = tk[i];
      d[6] = t_now - t_begin;
      d[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;
    }
#else
    er.template write_stats_grid<WM, WN, TH>(p, lds, ty, tx, by);
#endif
  }
}
#undef STAMP

// ------------------------------------------------------------------------------ weight packing
// per-output-channel scale: 2^k with max |w| * 2^k in [2^14, 2^15); oscale[c] = 2^-k (1 for padded / all-zero channels)
__global__ __launch_bounds__(64) void f16x3_wscale_kernel(const float* __restrict__ w, float* __restrict__ oscale,
                                                           int Cout, int K) {
  const int o = blockIdx.x;
  float m = 0.f;
  if (o < Cout)
    for (int i = threadIdx.x; i < K; i += 64) m = fmaxf(m, fabsf(w[(size_t)o * K + i]));
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(15 - e) in [2^14, 2^15)
      s = ldexpf(1.f, min(max(e - 15, -100), 100));
    }
    oscale[o] = s;
  }
}

// fp16 element index: ((((((nt * nchunks + ch) * NTAPS + tap) * 2 + nh) * 2 + nb) * 2 + plane) * 64 + lane) * 8 + j
//   -> plane (g1, g2) of w[o = nt*64 + nh*32 + nb*16 + (lane & 15)][c = chunk channel (lane >> 4)*8 + j][tap] * 2^k
// s2d: w is the 4x4 / stride-2 weight [Cout][C0][4][4]; the packed conv is 2x2 over 4*C0 channels ordered (py, px, c):
//      W2[o][(py, px, c)][dy][dx] = w[o][c][2*dy + py][2*dx + px]   (nch0 = 4 * C0 / 32 chunks)
__global__ void pack_f16x3_weight_kernel(const float* __restrict__ w, const float* __restrict__ oscale,
                                         _Float16* __restrict__ wp, int Cout, int C0, int C1, int NTAPS, int nch0,
                                         int nch1, int64_t total, int s2d, int pk) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t r = idx;
  const int j = r % 8;
  r /= 8;
  const int lane = r % 64;
  r /= 64;
  const int plane = r % 2;
  r /= 2;
  const int nb = r % 2;
  r /= 2;
  const int nh = r % 2;
  r /= 2;
  const int tap = r % NTAPS;
  r /= NTAPS;
  const int ch = r % (nch0 + nch1);
  const int nt = r / (nch0 + nch1);
  const int o = nt * 64 + nh * 32 + nb * 16 + (lane & 15);
  const int k = (lane >> 4) * 8 + j;
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
  float ws = 0.f;
  if (s2d) {
    const int nchc = C0 / KC, par = ch / nchc;
    const int cr = (ch - par * nchc) * KC + k;  // real input channel
    const int ky = 2 * (tap >> 1) + (par >> 1), kx = 2 * (tap & 1) + (par & 1);
    if (o < Cout) ws = w[((size_t)o * C0 + cr) * 16 + ky * 4 + kx] / oscale[o];
  } else if (pk) {  // 7x7, C0 <= 16: K slot k = (tap parity) * 16 + channel of tap pair `tap`
    const int tt = 2 * tap + (k >> 4), cc = k & 15;
    if (o < Cout && cc < C0 && tt < 49) ws = w[((size_t)o * C0 + cc) * 49 + tt] / oscale[o];
  } else if (ok && o < Cout) {
    ws = w[((size_t)o * (C0 + C1) + c) * NTAPS + tap] / oscale[o];  // exact: a power of two
  }
  const _Float16 g1 = (_Float16)ws;
  const _Float16 g2 = (_Float16)(ws - (float)g1);
  wp[idx] = plane == 0 ? g1 : g2;
}

// ------------------------------------------------------------------------------ many weights at once (training re-pack)
// dmh_pack_conv_weights_multi: the three phases of ws_standardize -> f16x3_wscale -> pack_f16x3_weight as table-driven
// kernels over up to PM_MAX jobs per launch (the table travels as a kernel argument: HIP-graph capturable, no device table
// to keep in step with the caller's buffers).  Plain stride-1 1x1 / 3x3 images only.
static int64_t f16x3_frag_floats(int Cout, int C0, int C1, int KH, int KW);
#define PM_MAX 32
struct PmTable {
  const float* src[PM_MAX];
  float* ws[PM_MAX];
  float* wpack[PM_MAX];
  int Cout[PM_MAX], C0[PM_MAX], C1[PM_MAX], T[PM_MAX], tr[PM_MAX];
  int blk0[PM_MAX + 1];
  int count;
};
__device__ __forceinline__ int pm_find(const PmTable& t, int blk) {
  int lo = 0, hi = t.count;  // blk0[lo] <= blk < blk0[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.blk0[mid] <= blk) lo = mid; else hi = mid;
  }
  return lo;
}
// element (o, c, tap) of the weight an image is made from; transposed: the data-gradient conv's weight, read from the
// forward weight [C][Cout][T] with the taps flipped
__device__ __forceinline__ float pm_elem(const float* w, int o, int c, int tap, int Cout, int Cin, int T, int tr) {
  return tr ? w[((size_t)c * Cout + o) * T + (T - 1 - tap)] : w[((size_t)o * Cin + c) * T + tap];
}
// phase 1: one block per (job, source output channel) — ws_standardize_kernel of conv.hip, same order of operations
__global__ __launch_bounds__(256) void pm_ws_kernel(PmTable t, float eps) {
  __shared__ float red[8];
  const int ti = pm_find(t, blockIdx.x), o = blockIdx.x - t.blk0[ti];
  const int K = (t.tr[ti] ? t.Cout[ti] : t.C0[ti] + t.C1[ti]) * t.T[ti];   // (a job's ws is in the SOURCE layout)
  const float* wr = t.src[ti] + (size_t)o * K;
  float* out = t.ws[ti] + (size_t)o * K;
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
  for (int i = threadIdx.x; i < K; i += 256) out[i] = (wr[i] - mean) * rstd;
}
// phase 2: one 64-thread block per (job, padded output channel) — f16x3_wscale_kernel
__global__ __launch_bounds__(64) void pm_wscale_kernel(PmTable t) {
  const int ti = pm_find(t, blockIdx.x), o = blockIdx.x - t.blk0[ti];
  const int Cout = t.Cout[ti], Cin = t.C0[ti] + t.C1[ti], T = t.T[ti], tr = t.tr[ti], K = Cin * T;
  const float* w = t.ws[ti] ? t.ws[ti] : t.src[ti];
  float* oscale = t.wpack[ti] + f16x3_frag_floats_dev(Cout, t.C0[ti], t.C1[ti], T);
  float m = 0.f;
  if (o < Cout)
    for (int i = threadIdx.x; i < K; i += 64) m = fmaxf(m, fabsf(pm_elem(w, o, i / T, i % T, Cout, Cin, T, tr)));
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) {
    float sc = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);
      sc = ldexpf(1.f, min(max(e - 15, -100), 100));
    }
    oscale[o] = sc;
  }
}
// phase 3: 256 fp16 elements of an image per block — pack_f16x3_weight_kernel (plain form)
__global__ __launch_bounds__(256) void pm_pack_kernel(PmTable t) {
  const int ti = pm_find(t, blockIdx.x);
  const int Cout = t.Cout[ti], C0 = t.C0[ti], C1 = t.C1[ti], T = t.T[ti], tr = t.tr[ti];
  const int nch0 = (C0 + KC - 1) / KC, nch1 = (C1 + KC - 1) / KC;
  const int64_t frag = f16x3_frag_floats_dev(Cout, C0, C1, T);
  const int64_t total = frag * 2;
  const float* w = t.ws[ti] ? t.ws[ti] : t.src[ti];
  const float* oscale = t.wpack[ti] + frag;
  _Float16* wp = reinterpret_cast<_Float16*>(t.wpack[ti]);
  // (a block covers PM_PACK_PER_BLOCK consecutive elements)
  const int64_t base = (int64_t)(blockIdx.x - t.blk0[ti]) * 2048;
  for (int64_t idx = base + threadIdx.x; idx < min(base + 2048, total); idx += 256) {
    int64_t r = idx;
    const int j = r % 8;
    r /= 8;
    const int lane = r % 64;
    r /= 64;
    const int plane = r % 2;
    r /= 2;
    const int nb = r % 2;
    r /= 2;
    const int nh = r % 2;
    r /= 2;
    const int tap = r % T;
    r /= T;
    const int ch = r % (nch0 + nch1);
    const int nt = r / (nch0 + nch1);
    const int o = nt * 64 + nh * 32 + nb * 16 + (lane & 15);
    const int k = (lane >> 4) * 8 + j;
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
    float ws = 0.f;
    if (ok && o < Cout) ws = pm_elem(w, o, c, tap, Cout, C0 + C1, T, tr) / oscale[o];  // exact: a power of two
    const _Float16 g1 = (_Float16)ws;
    const _Float16 g2 = (_Float16)(ws - (float)g1);
    wp[idx] = plane == 0 ? g1 : g2;
  }
}

int dmh_f16x3_pack_multi(const DmhPackJob* jobs, int njobs, float eps, hipStream_t st) {
  for (int phase = 0; phase < 3; ++phase) {
    for (int i0 = 0; i0 < njobs;) {
      PmTable t;
      int k = 0, blocks = 0;
      for (; k < PM_MAX && i0 < njobs; ++i0) {
        const DmhPackJob& j = jobs[i0];
        if (phase == 0 && !j.ws) continue;
        const int T = j.KH * j.KH;
        t.src[k] = j.src;
        t.ws[k] = j.ws;
        t.wpack[k] = j.wpack;
        t.Cout[k] = j.Cout;
        t.C0[k] = j.C0;
        t.C1[k] = j.C1;
        t.T[k] = T;
        t.tr[k] = j.transposed;
        t.blk0[k] = blocks;
        if (phase == 0) blocks += j.transposed ? j.C0 + j.C1 : j.Cout;                 // source output channels
        else if (phase == 1) blocks += cdiv(j.Cout, 64) * 64;
        else blocks += (int)cdiv64(f16x3_frag_floats(j.Cout, j.C0, j.C1, j.KH, j.KH) * 2, 2048);
        ++k;
      }
      if (k == 0) continue;
      t.blk0[k] = blocks;
      t.count = k;
      if (phase == 0) hipLaunchKernelGGL(pm_ws_kernel, dim3(blocks), dim3(256), 0, st, t, eps);
      else if (phase == 1) hipLaunchKernelGGL(pm_wscale_kernel, dim3(blocks), dim3(64), 0, st, t);
      else hipLaunchKernelGGL(pm_pack_kernel, dim3(blocks), dim3(256), 0, st, t);
      DMH_CHECK_LAUNCH("dmh_pack_conv_weights_multi");
    }
  }
  return DMH_OK;
}

// the 7x7 init conv with few input channels runs two taps per K slice (UPS == 4)
static bool f16x3_pk(int C0, int C1, int KH) { return KH == 7 && C1 == 0 && C0 <= 16; }

static int64_t f16x3_frag_floats(int Cout, int C0, int C1, int KH, int KW) {
  if (KH == 4) return (int64_t)cdiv(Cout, 64) * (4 * C0 / KC) * 4 * 2 * STEP_U4 * 4;  // as 2x2 over 4*C0 channels
  if (f16x3_pk(C0, C1, KH)) return (int64_t)cdiv(Cout, 64) * 25 * 2 * STEP_U4 * 4;      // one chunk, 25 tap pairs
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * KH * KW * 2 * STEP_U4 * 4;
}

int64_t dmh_f16x3_pack_floats(int Cout, int C0, int C1, int KH, int KW) {
  return f16x3_frag_floats(Cout, C0, C1, KH, KW) + (int64_t)cdiv(Cout, 64) * 64;  // + the per-channel 2^-k
}

int dmh_f16x3_pack(const float* w, float* wpack, int Cout, int C0, int C1, int KH, int KW, hipStream_t st) {
  const bool s2d = KH == 4;
  const int nch0 = s2d ? 4 * C0 / KC : cdiv(C0, KC), nch1 = s2d ? 0 : cdiv(C1, KC);
  const int64_t frag = f16x3_frag_floats(Cout, C0, C1, KH, KW);
  float* oscale = wpack + frag;
  hipLaunchKernelGGL(f16x3_wscale_kernel, dim3(cdiv(Cout, 64) * 64), dim3(64), 0, st, w, oscale, Cout,
                     (C0 + C1) * KH * KW);
  const int64_t total = frag * 2;  // fp16 elements
  const bool pk = f16x3_pk(C0, C1, KH);
  hipLaunchKernelGGL(pack_f16x3_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w, oscale,
                     reinterpret_cast<_Float16*>(wpack), Cout, C0, C1, s2d ? 4 : (pk ? 25 : KH * KW), nch0, nch1, total,
                     s2d ? 1 : 0, pk ? 1 : 0);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight(f16x3)");
  return DMH_OK;
}

template <int KH, int KW, int S, int UPS, int TH, int TW, int WM, int WN>
static int launch_f16x3(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = F16Cfg<KH, KW, S, UPS, TH, TW, WM, WN>;
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, TW);
  if (UPS == 2) a.nch0 = 4 * a.C0 / KC;  // (parity, 32-channel) chunks of the space-to-depth view
  a.oscale = d->wpack + f16x3_frag_floats(d->Cout, a.C0, a.C1, UPS == 2 ? 4 : KH, UPS == 2 ? 4 : KW);
#ifdef DMH_STAMPS
  if (const char* e = getenv("DMH_WINO_ABLATE")) a.ablate = atoi(e);
#endif
  auto kern = conv_f16x3_kernel<KH, KW, S, UPS, TH, TW, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Cfg::LDS_BYTES);
    DMH_REQUIRE(e == hipSuccess, "dmh_conv2d: cannot raise the LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  // round 6: at the levels with Hout <= 32 the cout tiles of one pixel tile are dealt to the SAME XCD (a.xcd = 2: the re-deal
  // walks the cout tile innermost; a function of the layer shape only, results bitwise unchanged).  Measured with a tight
  // A/B (10 alternating rounds, standard error 0.05 %): +0.24 % / +0.29 % / +0.29 % images/s on three boxes, all of it from
  // the launches with four cout tiles (512-wide @16^2: one under-filled round of workgroups, where every XCD now fetches its
  // input tiles once instead of four XCDs each), although their FETCH_SIZE goes UP (weights re-streamed per XCD: from the
  // Infinity Cache); docs/EXPERIMENTS.md R6.1.  DMH_CONV_XCD_DEEP=0 restores the order of rounds 1-5 (cout tile outermost).
  static const int xcd_deep = [] {
    const char* e = getenv("DMH_CONV_XCD_DEEP");
    return e ? atoi(e) : 32;
  }();
  static const int xcd_deep_maxy = [] {   // (development knob: only launches with at most this many cout tiles)
    const char* e = getenv("DMH_CONV_XCD_DEEP_MAXY");
    return e ? atoi(e) : 1 << 20;
  }();
  if (a.xcd == 1 && xcd_deep > 0 && Hout <= xcd_deep && cdiv(a.Cout, 64 * WN) > 1 && cdiv(a.Cout, 64 * WN) <= xcd_deep_maxy) a.xcd = 2;
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64 * WN));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d(f16x3)");
  return DMH_OK;
}

// ---- sub-pixel form of Upsample + conv3x3 (UPS == 3)
// wv[(par*Cout + o)][c][a][b] = sum of w[o][c][ky][kx] over the taps that fall on low-resolution source pixel (a, b) of
// parity par = (dy, dx):  dy = 0: a = 0 <- ky 0, a = 1 <- ky 1, 2;   dy = 1: a = 0 <- ky 0, 1, a = 1 <- ky 2  (same in x)
__global__ void subpixel_weight_kernel(const float* __restrict__ w, float* __restrict__ wv, int Cout, int Cin) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)4 * Cout * Cin * 4;
  if (idx >= total) return;
  const int tb = idx & 1, ta = (idx >> 1) & 1;
  int64_t r = idx >> 2;
  const int c = r % Cin;
  r /= Cin;
  const int o = r % Cout, par = r / Cout;
  const int dy = par >> 1, dx = par & 1;
  const int ky0 = dy == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), ky1 = dy == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
  const int kx0 = dx == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), kx1 = dx == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
  const float* wk = w + ((size_t)o * Cin + c) * 9;
  float s = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) s += wk[ky * 3 + kx];
  wv[idx] = s;
}

int64_t dmh_f16x3_up2_pack_floats(int Cout, int C0) {
  return dmh_f16x3_pack_floats(4 * Cout, C0, 0, 2, 2) + (int64_t)16 * Cout * C0;  // + the summed weights (pack-time scratch)
}

int dmh_f16x3_up2_pack(const float* w, float* wpack, int Cout, int C0, hipStream_t st) {
  float* wv = wpack + dmh_f16x3_pack_floats(4 * Cout, C0, 0, 2, 2);
  const int64_t total = (int64_t)16 * Cout * C0;
  hipLaunchKernelGGL(subpixel_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w, wv, Cout, C0);
  return dmh_f16x3_pack(wv, wpack, 4 * Cout, C0, 0, 2, 2, st);
}

int dmh_f16x3_launch_up2(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = F16Cfg<2, 2, 1, 3, 4, 16, 4, 1>;
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, 4, 16);
  a.tilesX = cdiv(d->Win, 16);  // tiles live on the low-resolution grid
  a.tilesY = cdiv(d->Hin, 4);
  a.oscale = d->wpack + f16x3_frag_floats(4 * d->Cout, a.C0, 0, 2, 2);
  auto kern = conv_f16x3_kernel<2, 2, 1, 3, 4, 16, 4, 1>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Cfg::LDS_BYTES);
    DMH_REQUIRE(e == hipSuccess, "dmh_conv2d: cannot raise the LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d(f16x3 sub-pixel)");
  return DMH_OK;
}

// 3x3 / 1x1 stride 1 (3x3 optionally behind a nearest x2 upsample).  Cout a multiple of 128: 8x16 pixels x 128
// channels per workgroup (half the staging per flop); otherwise 16x16 pixels x 64 channels.
int dmh_f16x3_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  const bool wide = d->Cout % 128 == 0;
  if (d->KH == 4) {  // Downsample: 2x2 over the shifted space-to-depth view
    return wide ? launch_f16x3<2, 2, 1, 2, 8, 16, 2, 2>(d, Hout, Wout, st)
                : launch_f16x3<2, 2, 1, 2, 16, 16, 4, 1>(d, Hout, Wout, st);
  }
  if (d->KH == 7) {  // init conv (CFG:333): few input channels — two taps per 32-channel K slice when they fit
    if (f16x3_pk(d->C0, d->src1 ? d->C1 : 0, 7)) return launch_f16x3<7, 7, 1, 4, 16, 16, 4, 1>(d, Hout, Wout, st);
    return launch_f16x3<7, 7, 1, 0, 16, 16, 4, 1>(d, Hout, Wout, st);
  }
  if (d->KH == 1) {
    static int wide1 = -1;  // development knob
    if (wide1 < 0) {
      const char* e = getenv("DMH_F16_WIDE1");
      wide1 = e ? atoi(e) : 1;
    }
    return (wide && wide1) ? launch_f16x3<1, 1, 1, 0, 8, 16, 2, 2>(d, Hout, Wout, st)
                : launch_f16x3<1, 1, 1, 0, 16, 16, 4, 1>(d, Hout, Wout, st);
  }
  if (d->upsample2) {
    return wide ? launch_f16x3<3, 3, 1, 1, 8, 16, 2, 2>(d, Hout, Wout, st)
                : launch_f16x3<3, 3, 1, 1, 16, 16, 4, 1>(d, Hout, Wout, st);
  }
  static int wide3 = -1;  // development knob (DMH_F16_WIDE3=0: the 16x16x64 layout for every plain 3x3)
  if (wide3 < 0) {
    const char* e = getenv("DMH_F16_WIDE3");
    wide3 = e ? atoi(e) : 1;
  }
  if (wide && wide3) return launch_f16x3<3, 3, 1, 0, 8, 16, 2, 2>(d, Hout, Wout, st);
  // (the 16 x 16 layout keeps a load's source pixel relative to the tile in 16 bits: PACKPW in the kernel)
  DMH_REQUIRE(17 * (int64_t)d->Win + 17 < 65536, "dmh_conv2d: 3x3 with Cout %% 128 != 0 supports Win <= 3853 (got %d)", d->Win);
  return launch_f16x3<3, 3, 1, 0, 16, 16, 4, 1>(d, Hout, Wout, st);
}
