// K1, Winograd form on the fp16 matrix cores — 3x3 / stride-1 convolution as F(2x2, 3x3) whose Winograd-domain products
// run as block-scaled fp16 pieces ("f16x3": three v_mfma_f32_32x32x16_f16 per product block, fp32 accumulation; the
// arithmetic of conv_f16x3.hip applied to the transformed operands).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A            16 products per 2x2 outputs instead of 36
//
//   activations  V = B^T d B in fp32 (adds only), then  h1 = fp16(V * s), h2 = fp16(V * s - h1):  s per (workgroup,
//                running over the chunks) from the largest |d| staged so far — |V| <= 4 max|d|, so s puts 4 max|d| below
//                2^15; an element keeps 22 significant bits unless it is 2^16 below the block maximum (fp16 pieces are
//                floating point: the slack of the bound costs range, not precision)
//   weights      U = G g G^T in fp32 at pack time, * 2^k per output channel with 2.25 max|g| 2^k < 2^15, as g1 + g2
//   product      U V s 2^k = g1 h2 + g2 h1 + g1 h1 + O(2^-22)
//   output       y = (A^T M A) / s * 2^-k + bias
//
// Per workgroup: 8x16 output pixels (32 Winograd tiles) x 64 output channels; per 16-channel chunk
//   1. the 10x18 input halo goes to LDS in fp32 through the fused GroupNorm+SiLU prologue (its largest magnitude -> slot),
//   2. every thread transforms one (tile, channel quad, pair of Winograd rows): 12 LDS reads, 32 packed adds, splits its 32
//      results and writes them as the fp16 B fragments of the matrix phase (double buffered),
//   3. wave w multiplies the four positions of Winograd row w: M_pos[64 cout][32 tiles] — the WEIGHT fragment is the first
//      MFMA operand, so a lane holds 4 consecutive output channels of one tile; weights stream from L2 in fragment order.
// The steps of consecutive chunks are software pipelined, two barriers per chunk.
// Output transform: wave w contracts its row over nu in registers, the rows are exchanged through LDS (16 B pieces) and
// contracted over xi while the shared row epilogue (+bias, +residual, GroupNorm partials) reads them.
//
// Replaces: the 3x3 convolutions of Block / ResnetBlock CFG:128-170 (DMH_CONV3_VARIANT=10).
#include <stdlib.h>

#include "common.h"

#include "conv_args.h"

// Winograd 3x3 on fp16 pieces (conv_wino_f16x3.hip)
int64_t dmh_winof_pack_floats(int Cout, int C0, int C1);
int dmh_winof_pack(const float* w, float* wpack, int Cout, int C0, int C1, hipStream_t st);
int dmh_winof_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st);

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

namespace {
constexpr int TH = 8, TW = 16, KC = 16, IN_H = 10, IN_W = 18, IN_PIX = IN_H * IN_W, NT = 32;
constexpr int RAWP = 20;                         // raw tile pitch in floats
constexpr int RAW_FLOATS = IN_PIX * RAWP;        // 3600
constexpr int V_BYTES = 16 * 2 * NT * 32;        // 32768 per buffer: [pos][piece][tile][16 channels] fp16
constexpr int ZP = 68;                           // exchange pitch in floats
constexpr int Z_FLOATS = 4 * 2 * NT * ZP;        // 17408: [row xi][j][tile][cout]
constexpr int MAIN_BYTES = RAW_FLOATS * 4 + 2 * V_BYTES;      // 79936
static_assert(Z_FLOATS * 4 <= MAIN_BYTES, "the exchange buffer overlays the pipeline buffers");
constexpr int LDS_BYTES = MAIN_BYTES + 64;       // + [chunk parity][wave] block-maximum slots: two workgroups per CU
constexpr int NTHR = 512;
constexpr int NLOAD = (IN_PIX * 4 + NTHR - 1) / NTHR;  // halo float4 slots per thread per chunk
constexpr int POS_U4 = 4 * 64;                   // uint4 per position of packed U: 2 column blocks x 2 pieces x 64 lanes

struct f4 {  // a float4 as two packed pairs: + and - compile to v_pk_add_f32
  float2v lo, hi;
};
__device__ __forceinline__ f4 ldf4(const float* p) {
  const float4 t = ld4(p);
  return f4{float2v{t.x, t.y}, float2v{t.z, t.w}};
}
__device__ __forceinline__ f4 operator+(const f4& a, const f4& b) { return f4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ f4 operator-(const f4& a, const f4& b) { return f4{a.lo - b.lo, a.hi - b.hi}; }

struct True_ {
  static constexpr bool value = true;
};
struct False_ {
  static constexpr bool value = false;
};
}  // namespace

// Software pipeline, one stage per chunk c, two barriers per stage:
//   half 1:  input transform + split of chunk c+1: raw -> V[(c+1) & 1];   matrix(c) positions 0,1
//   half 2:  raw tile of chunk c+2 (registers -> prologue -> LDS), loads of chunk c+3;   matrix(c) positions 2,3
// Eight waves per workgroup: wave = (Winograd row xi, half of the 64 output channels), 64 accumulator registers each, so
// that two workgroups = sixteen waves share a CU (<= 128 registers): four waves per SIMD to cover each other's LDS / L2
// round trips, where the direct kernel (244 registers) has two.
template <int UPS>
__global__ __launch_bounds__(512, 4) void conv_wino_f16x3_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;
  unsigned char* V = reinterpret_cast<unsigned char*>(lds + RAW_FLOATS);
  unsigned* slot = reinterpret_cast<unsigned*>(V + 2 * V_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, nbw = wave >> 2;  // this wave's matrix work: Winograd row xi, output channels nbw*32 .. +31
  const int half = lane >> 5, l31 = lane & 31;

  int t = blockIdx.x, by = blockIdx.y;
  if (p.xcd) {  // XCD k walks the k-th contiguous run of tiles (see conv_f16x3.hip)
    const int nb = gridDim.x, total = nb * gridDim.y, lin = t + by * nb;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, local = lin >> 3;
    const int nl = xcd * q + min(xcd, r) + local;
    t = nl % nb;
    by = nl / nb;
  }
  const int tx0 = t % p.tilesX;
  t /= p.tilesX;
  const int ty0 = t % p.tilesY;
  const int b = t / p.tilesY;
  const int nt = by;
  const int n0 = nt * 64;
  const int tile_in_sample = ty0 * p.tilesX + tx0;
  const int oy0 = ty0 * TH, ox0 = tx0 * TW;

  // ---- chunk-invariant staging state: halo slots (pixel, channel quad c4) of this thread
  const int c4 = tid & 3;
  int poff[NLOAD];
  unsigned inside = 0;
  {
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int Hlim = UPS ? p.Hin * 2 : p.Hin;
    const int Wlim = UPS ? p.Win * 2 : p.Win;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int pix = (tid + i * NTHR) >> 2;
      const int pixc = pix < IN_PIX ? pix : IN_PIX - 1;
      const int yy = iy0 + pixc / IN_W, xx = ix0 + pixc % IN_W;
      const bool ok = pix < IN_PIX && yy >= 0 && yy < Hlim && xx >= 0 && xx < Wlim;
      const int yc = min(max(yy, 0), Hlim - 1), xc = min(max(xx, 0), Wlim - 1);
      const int sy = UPS ? (yc >> 1) : yc, sx = UPS ? (xc >> 1) : xc;
      poff[i] = (b * p.Hin + sy) * p.Win + sx;
      inside |= (ok ? 1u : 0u) << i;
    }
  }
  // this thread's transform item: tile, channel quad, Winograd row xt (uniform over a pair of waves)
  //   rows of d that enter row xt of B^T d:  0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3
  const int tq = tid & 3, ttile = (tid >> 2) & 31, xt = wave >> 1;
  const int ra_ = (xt == 0) ? 0 : (xt == 2 ? 2 : 1);
  const int rb_ = (xt == 0 || xt == 1) ? 2 : (xt == 2 ? 1 : 3);
  const float sgn = (xt == 1) ? 1.f : -1.f;
  const int rd_a = ((2 * (ttile >> 3) + ra_) * IN_W + 2 * (ttile & 7)) * RAWP + tq * 4;
  const int rd_b = ((2 * (ttile >> 3) + rb_) * IN_W + 2 * (ttile & 7)) * RAWP + tq * 4;
  const int wr_v = ((xt * 4) * 2 * NT + ttile) * 32 + tq * 8;   // position 4*xt + nu, piece 0; piece 1 is NT*32 further
  const int wr_raw0 = (tid >> 2) * RAWP + c4 * 4;  // slot i lives NTHR/4 pixels further

  floatx16 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int nchunks = p.nch0 + p.nch1;
  const int last = nchunks - 1;
  // packed U: [nt][chunk][pos 16][nb 2][piece 2][lane 64] x 16 B; this wave's positions are 4*xi .. 4*xi + 3, column block nbw
  const uint4* wu = reinterpret_cast<const uint4*>(p.wpack) +
                    ((size_t)__builtin_amdgcn_readfirstlane(nt) * nchunks * 16 + xi * 4) * POS_U4 + nbw * 128 + lane;
  const int nlin = nchunks * 4;  // (chunk, position-of-this-wave) pairs in matrix-phase order
  uint4 bq[2][2];
  auto load_b = [&](int buf, int lin) {
    const int l = lin < nlin ? lin : nlin - 1;
    const uint4* src = wu + ((size_t)(l >> 2) * 16 + (l & 3)) * POS_U4;
    bq[buf][0] = src[0];
    bq[buf][1] = src[64];
  };
  // B fragment of position 4*xi + q: lane (tile l31, K half) reads its 8 channels of each piece
  const unsigned char* va = V + ((xi * 4) * 2 * NT + l31) * 32 + half * 16;

  float4 v[NLOAD];
  float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue_chunk_loads = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    const float* src = s1 ? p.src1 : p.src0;
    const int Csrc = s1 ? p.C1 : p.C0;
    const int cbase = (s1 ? ch - p.nch0 : ch) * KC;
    const int cc = (cbase + c4 * 4 < Csrc) ? cbase + c4 * 4 : 0;  // clamped for the padded tail of the last chunk
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[i] = ld4(src + (size_t)poff[i] * Csrc + cc);
    if (p.in_coef != nullptr && !s1) {
      ca = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + cc);
      cb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + cc);
    }
  };
  // registers (chunk ch) -> (prologue SiLU(a*x+b) on every lane, padding cleared afterwards) -> raw LDS tile;
  // the wave's largest staged magnitude -> slot[ch & 1][wave]
  auto raw_write = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    const bool pro = (p.in_coef != nullptr) && !s1;
    const int Csrc = s1 ? p.C1 : p.C0;
    const bool cvalid = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4 < Csrc;
    const unsigned m = cvalid ? inside : 0u;
    if (pro) {
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        v[i].x = silu_fast(fmaf(ca.x, v[i].x, cb.x));
        v[i].y = silu_fast(fmaf(ca.y, v[i].y, cb.y));
        v[i].z = silu_fast(fmaf(ca.z, v[i].z, cb.z));
        v[i].w = silu_fast(fmaf(ca.w, v[i].w, cb.w));
      }
    }
    float mxf = 0.f;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const float k = ((m >> i) & 1u) ? 1.f : 0.f;   // padding stays exactly zero: it pads the ACTIVATED tensor
      const float4 x = make_float4(v[i].x * k, v[i].y * k, v[i].z * k, v[i].w * k);
      mxf = fmaxf(fmaxf(mxf, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
      if ((i + 1) * NTHR <= IN_PIX * 4 || ((tid + i * NTHR) >> 2) < IN_PIX) st4(raw + wr_raw0 + i * ((NTHR / 4) * RAWP), x);
    }
    const unsigned mx = wave_max_u32(__float_as_uint(mxf));
    if (lane == 0) slot[(ch & 1) * 8 + wave] = mx;
  };
  // block scale: running maximum over the chunks (the scale only ever shrinks: no overflow on rescale)
  int e_cur = 16, e_next = 16;  // biased exponents of the running maximum: chunk in the matrix phase / chunk being transformed
  // input transform V = B^T d B of the chunk in `raw`, split into fp16 pieces -> V[ch & 1]
  auto transform = [&](int ch) {
    const uint4 sa = *reinterpret_cast<const uint4*>(slot + (ch & 1) * 8), sb = *reinterpret_cast<const uint4*>(slot + (ch & 1) * 8 + 4);
    const unsigned bmx = max(max(max(sa.x, sa.y), max(sa.z, sa.w)), max(max(sb.x, sb.y), max(sb.z, sb.w)));
    const int e_ch = min(max((int)(__builtin_amdgcn_readfirstlane(bmx) >> 23), 16), 254);
    e_next = max(e_next, e_ch);
    const float sc = __uint_as_float((unsigned)(266 - e_next) << 23);  // 4 * largest |d| * sc < 2^15
    const float2v sg = float2v{sgn, sgn};
    f4 w[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f4 da = ldf4(raw + rd_a + c * RAWP), db = ldf4(raw + rd_b + c * RAWP);
      w[c].lo = da.lo + sg * db.lo;
      w[c].hi = da.hi + sg * db.hi;
    }
    unsigned char* vo = V + (ch & 1) * V_BYTES + wr_v;
    auto emit = [&](const f4& o, int nu) {
      uint2 h1, h2;
      dmh_split2(o.lo.x, o.lo.y, sc, h1.x, h2.x);
      dmh_split2(o.hi.x, o.hi.y, sc, h1.y, h2.y);
      *reinterpret_cast<uint2*>(vo + nu * (2 * NT * 32)) = h1;
      *reinterpret_cast<uint2*>(vo + nu * (2 * NT * 32) + NT * 32) = h2;
    };
    // V[xt][nu 0..3] = w0 - w2, w1 + w2, w2 - w1, w1 - w3
    emit(w[0] - w[2], 0);
    emit(w[1] + w[2], 1);
    emit(w[2] - w[1], 2);
    emit(w[1] - w[3], 3);
  };

  // position 4*xi + q of chunk c: M[32 cout][32 tiles] += U^T[cout][k] V^T[k][tiles], smallest terms first
#define DMH_POSITION(c, q)                                                                                         \
  {                                                                                                                \
    if (!(p.ablate & 16)) load_b(((q) & 1) ^ 1, (c) * 4 + (q) + 1); /* next position's weights */                  \
    __builtin_amdgcn_sched_barrier(0x38F);    /* memory loads stay up here */                                      \
    const unsigned char* vb = va + ((c) & 1) * V_BYTES + (q) * (2 * NT * 32);                                      \
    const half8 x1 = *reinterpret_cast<const half8*>(vb);                                                          \
    const half8 x2 = *reinterpret_cast<const half8*>(vb + NT * 32);                                                \
    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, bq[(q) & 1][0]), x2, acc[q], 0, 0, 0); \
    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, bq[(q) & 1][1]), x1, acc[q], 0, 0, 0); \
    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, bq[(q) & 1][0]), x1, acc[q], 0, 0, 0); \
  }

  auto stage = [&](int c, auto TR, auto RW) {
    if constexpr (decltype(TR)::value) if (!(p.ablate & 1)) transform(c + 1);
    if (!(p.ablate & 4)) {
    DMH_POSITION(c, 0)
    DMH_POSITION(c, 1)
    }
    __syncthreads();  // raw is free again (and V[(c+1)&1] is complete)
    if constexpr (decltype(RW)::value) if (!(p.ablate & 2)) {
      raw_write(c + 2);
      issue_chunk_loads(c + 3 < nchunks ? c + 3 : last);
    }
    if (!(p.ablate & 4)) {
    DMH_POSITION(c, 2)
    DMH_POSITION(c, 3)
    }
    if (e_next != e_cur) {  // the next chunk was split under a smaller scale: bring the sums to it
      const int fe = 127 + e_cur - e_next;
      const float f = fe > 0 ? __uint_as_float((unsigned)fe << 23) : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] *= f;
      e_cur = e_next;
    }
    __syncthreads();  // raw of chunk c+2 published; V[c & 1] is free
  };

  // ---- pipeline fill
  issue_chunk_loads(0);
  load_b(0, 0);
  raw_write(0);
  __syncthreads();
  issue_chunk_loads(1 < nchunks ? 1 : last);
  transform(0);
  e_cur = e_next;
  __syncthreads();
  if (nchunks > 1) raw_write(1);
  issue_chunk_loads(2 < nchunks ? 2 : last);
  __syncthreads();

  for (int c = 0; c < nchunks - 2; ++c) stage(c, True_{}, True_{});
  if (nchunks >= 2) stage(nchunks - 2, True_{}, False_{});
  stage(nchunks - 1, False_{}, False_{});
#undef DMH_POSITION

  // ---- output transform Y = A^T M A.  This wave holds row xi of M for 32 channels: contract over nu in registers,
  //      Z[xi][j] = sum_nu A[nu][j] M[xi][nu], exchange rows through LDS, contract over xi while reading.
  //      C/D layout of the 32x32 block: column = tile l31, rows (r & 3) + 8 * (r >> 2) + 4 * half = output channel
  float* Z = lds;  // (the last stage ended on a barrier: LDS is free)
  if (p.ablate & 8) {
    if (acc[0][0] + acc[1][3] + acc[2][5] + acc[3][7] == 12345.f) p.out[0] = 0.f;  // keep the matrix work alive
    return;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 z0, z1;
    z0.x = acc[0][g * 4 + 0] + acc[1][g * 4 + 0] + acc[2][g * 4 + 0];
    z0.y = acc[0][g * 4 + 1] + acc[1][g * 4 + 1] + acc[2][g * 4 + 1];
    z0.z = acc[0][g * 4 + 2] + acc[1][g * 4 + 2] + acc[2][g * 4 + 2];
    z0.w = acc[0][g * 4 + 3] + acc[1][g * 4 + 3] + acc[2][g * 4 + 3];
    z1.x = acc[1][g * 4 + 0] - acc[2][g * 4 + 0] - acc[3][g * 4 + 0];
    z1.y = acc[1][g * 4 + 1] - acc[2][g * 4 + 1] - acc[3][g * 4 + 1];
    z1.z = acc[1][g * 4 + 2] - acc[2][g * 4 + 2] - acc[3][g * 4 + 2];
    z1.w = acc[1][g * 4 + 3] - acc[2][g * 4 + 3] - acc[3][g * 4 + 3];
    float* z = Z + ((xi * 2) * NT + l31) * ZP + nbw * 32 + 8 * g + 4 * half;
    st4(z, z0);
    st4(z + NT * ZP, z1);
  }
  EpilogueRows er(p, b, n0);
  {
    const float inv_s = __uint_as_float((unsigned)(e_cur - 12) << 23);  // 1 / sc rides on the per-channel 2^-k
    er.osc.x *= inv_s;
    er.osc.y *= inv_s;
    er.osc.z *= inv_s;
    er.osc.w *= inv_s;
  }
  __syncthreads();
  {
    const int c4e = er.c4;
    const float* Zc = Z + c4e * 4;
    er.template store_rows_fn<TW, 4>(
        p,
        [Zc, wave](int rr) {
          const int row = wave * 16 + rr;  // pixel (row >> 4, row & 15) of the 8x16 region: this wave owns tile row `wave`
          const int py = row >> 4, px = row & 15;
          const int tile = (py >> 1) * 8 + (px >> 1);
          const float* z = Zc + ((px & 1) * NT + tile) * ZP;
          const float4 z1 = ld4(z + 1 * (2 * NT * ZP)), z2 = ld4(z + 2 * (2 * NT * ZP));
          float4 o;
          if ((py & 1) == 0) {
            const float4 z0 = ld4(z);
            o = make_float4(z0.x + z1.x + z2.x, z0.y + z1.y + z2.y, z0.z + z1.z + z2.z, z0.w + z1.w + z2.w);
          } else {
            const float4 z3 = ld4(z + 3 * (2 * NT * ZP));
            o = make_float4(z1.x - z2.x - z3.x, z1.y - z2.y - z3.y, z1.z - z2.z - z3.z, z1.w - z2.w - z3.w);
          }
          return o;
        },
        wave * 16, oy0, ox0);
  }
  // GroupNorm partials of the 8x16 tile: lanes -> waves -> stats[b][tile][Cout][2] (fixed order, no atomics)
  if (p.stats) {
    er.s1.x = rows_sum(er.s1.x);
    er.s1.y = rows_sum(er.s1.y);
    er.s1.z = rows_sum(er.s1.z);
    er.s1.w = rows_sum(er.s1.w);
    er.s2.x = rows_sum(er.s2.x);
    er.s2.y = rows_sum(er.s2.y);
    er.s2.z = rows_sum(er.s2.z);
    er.s2.w = rows_sum(er.s2.w);
    __syncthreads();  // every wave is done with the exchange buffer: reuse LDS as the cross-wave scratch
    float* red = lds;
    if (lane < 16) {
      float* q = red + (wave * 64 + er.c4 * 4) * 2;
      q[0] = er.s1.x;
      q[1] = er.s2.x;
      q[2] = er.s1.y;
      q[3] = er.s2.y;
      q[4] = er.s1.z;
      q[5] = er.s2.z;
      q[6] = er.s1.w;
      q[7] = er.s2.w;
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < p.Cout) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
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
// per-output-channel scale: 2^k with 4 max |g| 2^k in [2^14, 2^15) (|U| <= 2.25 max|g|); oscale[c] = 2^-k
__global__ __launch_bounds__(64) void winof_wscale_kernel(const float* __restrict__ w, float* __restrict__ oscale, int Cout,
                                                          int K) {
  const int o = blockIdx.x;
  float m = 0.f;
  if (o < Cout)
    for (int i = threadIdx.x; i < K; i += 64) m = fmaxf(m, fabsf(w[(size_t)o * K + i]));
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) {
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(13 - e) in [2^12, 2^13)
      s = ldexpf(1.f, min(max(e - 13, -100), 100));
    }
    oscale[o] = s;
  }
}

// transformed weights U = G g G^T (fp32), * 2^k, split into two fp16 pieces, in fragment-major order
//   fp16 index = ((((((nt * nchunks + ch) * 16 + pos) * 2 + nb) * 2 + piece) * 64 + lane) * 8 + j
//   -> piece of U_pos[k = (lane >> 5) * 8 + j][cout = nt*64 + nb*32 + (lane & 31)]
__global__ void pack_winof_weight_kernel(const float* __restrict__ w, const float* __restrict__ oscale,
                                         _Float16* __restrict__ wp, int Cout, int C0, int C1, int nch0, int nch1,
                                         int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t r = idx;
  const int j = r % 8;
  r /= 8;
  const int lane = r % 64;
  r /= 64;
  const int piece = r % 2;
  r /= 2;
  const int nb = r % 2;
  r /= 2;
  const int pos = r % 16;
  r /= 16;
  const int ch = r % (nch0 + nch1);
  const int nt = r / (nch0 + nch1);
  const int o = nt * 64 + nb * 32 + (lane & 31);
  const int k = (lane >> 5) * 8 + j;
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
  if (ok && o < Cout) {
    const float* g = w + ((size_t)o * (C0 + C1) + c) * 9;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const int xi = pos >> 2, nu = pos & 3;
    float acc = 0.f;
    for (int a = 0; a < 3; ++a) {
      float row = 0.f;
      for (int bb = 0; bb < 3; ++bb) row = fmaf(g[a * 3 + bb], G[nu][bb], row);
      acc = fmaf(G[xi][a], row, acc);
    }
    val = acc / oscale[o];  // exact: a power of two
  }
  const _Float16 g1 = (_Float16)val;
  const _Float16 g2 = (_Float16)(val - (float)g1);
  wp[idx] = piece == 0 ? g1 : g2;
}

static int64_t winof_frag_floats(int Cout, int C0, int C1) {
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * 16 * POS_U4 * 4;  // uint4 = 4 floats
}

int64_t dmh_winof_pack_floats(int Cout, int C0, int C1) {
  return winof_frag_floats(Cout, C0, C1) + (int64_t)cdiv(Cout, 64) * 64;  // + the per-channel 2^-k
}

int dmh_winof_pack(const float* w, float* wpack, int Cout, int C0, int C1, hipStream_t st) {
  const int nch0 = cdiv(C0, KC), nch1 = cdiv(C1, KC);
  const int64_t frag = winof_frag_floats(Cout, C0, C1);
  float* oscale = wpack + frag;
  hipLaunchKernelGGL(winof_wscale_kernel, dim3(cdiv(Cout, 64) * 64), dim3(64), 0, st, w, oscale, Cout, (C0 + C1) * 9);
  const int64_t total = frag * 2;  // fp16 elements
  hipLaunchKernelGGL(pack_winof_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w, oscale,
                     reinterpret_cast<_Float16*>(wpack), Cout, C0, C1, nch0, nch1, total);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight(winograd f16x3)");
  return DMH_OK;
}

int dmh_winof_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, TW);
  a.oscale = d->wpack + winof_frag_floats(d->Cout, a.C0, a.C1);
  if (const char* e = getenv("DMH_WINOF_ABL")) a.ablate = atoi(e);
  static bool attr = false;
  if (!attr) {
    hipError_t e0 = hipFuncSetAttribute((const void*)conv_wino_f16x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        LDS_BYTES);
    hipError_t e1 = hipFuncSetAttribute((const void*)conv_wino_f16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        LDS_BYTES);
    DMH_REQUIRE(e0 == hipSuccess && e1 == hipSuccess, "dmh_conv2d: cannot raise the LDS limit");
    attr = true;
  }
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
  if (d->upsample2)
    hipLaunchKernelGGL((conv_wino_f16x3_kernel<1>), grid, dim3(NTHR), LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL((conv_wino_f16x3_kernel<0>), grid, dim3(NTHR), LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d(winograd f16x3)");
  return DMH_OK;
}
