// K1 fast path — 3x3 / stride-1 convolution by Winograd F(2x2, 3x3) with the Winograd-domain products on the
// bf16 matrix cores, every fp32 operand carried as three bf16 pieces ("bf16x3 split", six cross terms, fp32
// accumulation: the split is exact and the dropped terms are <= 2^-23 relative — see conv_bf16x3.hip).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A
//
// Why not the fp32 MFMA of conv_wino.hip: on gfx950 the fp32 MFMA executes on the vector ALUs, so the input
// transform, the prologue and the epilogue can never hide behind it (tools/micro/coexec.hip, tools/wino_ablate.py:
// the phases add up exactly).  The bf16 MFMAs run on the matrix cores proper, beside the VALU work of the other
// resident workgroup; the kernel becomes bound by its vector work (staging + transforms), the matrix phase
// (48 MFMAs per wave per chunk) hides under it.
//
// Per workgroup: 8x16 output pixels (32 Winograd tiles) x 64 output channels; per 16-channel chunk
//   1. the 10x18 input halo is staged to LDS (fp32) through the fused GroupNorm+SiLU prologue,
//   2. all 256 threads transform it, V = B^T d B -> V[16 positions][k quarter][32 tiles][4] (fp32, double buffered),
//   3. wave w multiplies the four positions of Winograd row w: M_pos[32 tiles][64 cout] on
//      v_mfma_f32_32x32x16_bf16; it splits its A fragment into bf16 pieces at the load, the pre-transformed,
//      pre-split weights U stream from L2 in fragment order.
// The three steps of consecutive chunks are software pipelined (see the kernel).
// Output transform: wave w contracts its row over nu in registers, the rows are exchanged through LDS and
// contracted over xi while the shared float4 row epilogue (+bias, +residual, GroupNorm partials) reads them.
//
// Replaces: the 3x3 convolutions of Block / ResnetBlock CFG:128-170 and the Upsample conv CFG:106-107.
#include <stdlib.h>

#include "conv_args.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

namespace {
constexpr int TH = 8, TW = 16, KC = 16, IN_H = 10, IN_W = 18, IN_PIX = IN_H * IN_W, NT = 32;
constexpr int RAWP = 20;                         // raw tile pitch in floats
constexpr int RAW_FLOATS = IN_PIX * RAWP;        // 3600
constexpr int V_FLOATS = 16 * 4 * NT * 4;        // 8192 per buffer: [pos][k quarter][tile][4] fp32
constexpr int ZP = 68;                           // exchange pitch in floats
constexpr int Z_FLOATS = 4 * 2 * NT * ZP;        // 17408: [row xi][j][tile][cout]
constexpr int MAIN_FLOATS = RAW_FLOATS + 2 * V_FLOATS;
constexpr int LDS_FLOATS = MAIN_FLOATS > Z_FLOATS ? MAIN_FLOATS : Z_FLOATS;  // 79936 B: two workgroups per CU
constexpr int LDS_BYTES = LDS_FLOATS * 4;
constexpr int NLOAD = (IN_PIX * 4 + 255) / 256;  // halo float4 slots per thread per chunk
constexpr int POS_U4 = 6 * 64;                   // uint4 per position of packed U: 2 column blocks x 3 pieces x 64 lanes

struct f4 {  // a float4 as two packed pairs: + and - compile to v_pk_add_f32
  float2v lo, hi;
};
__device__ __forceinline__ f4 ldf4(const float* p) {
  const float4 t = ld4(p);
  return f4{float2v{t.x, t.y}, float2v{t.z, t.w}};
}
__device__ __forceinline__ void stf4(float* p, const f4& v) { st4(p, make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y)); }
__device__ __forceinline__ f4 operator+(const f4& a, const f4& b) { return f4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ f4 operator-(const f4& a, const f4& b) { return f4{a.lo - b.lo, a.hi - b.hi}; }

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {  // top halves of (a, b) -> two bf16, truncation
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// eight floats (one lane's K slice of an A fragment) -> three bf16x8 pieces; both subtractions are exact
__device__ __forceinline__ void split8(const float4& f0, const float4& f1, bf16x8 (&a)[3]) {
  const float x[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
  float r[8], s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r[i] = x[i] - trunc_bf16(x[i]);
    s[i] = r[i] - trunc_bf16(r[i]);
  }
  a[0] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(x[0], x[1]), pack_hi16(x[2], x[3]), pack_hi16(x[4], x[5]),
                                               pack_hi16(x[6], x[7])));
  a[1] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(r[0], r[1]), pack_hi16(r[2], r[3]), pack_hi16(r[4], r[5]),
                                               pack_hi16(r[6], r[7])));
  a[2] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(s[0], s[1]), pack_hi16(s[2], s[3]), pack_hi16(s[4], s[5]),
                                               pack_hi16(s[6], s[7])));
}

struct True_ {
  static constexpr bool value = true;
};
struct False_ {
  static constexpr bool value = false;
};
}  // namespace

// Software pipeline, one stage per chunk c, two barriers per stage, everything between two barriers in ONE
// instruction stream so that the bf16 MFMAs (asynchronous once issued) run beside the vector work:
//   half 1:  matrix(c) positions 0,1   +  input transform of chunk c+1: raw -> V[(c+1) & 1]
//   half 2:  matrix(c) positions 2,3   +  raw tile of chunk c+2 (registers -> prologue -> LDS), loads of chunk c+3
// V holds fp32; the wave that multiplies a position splits its A fragment into bf16 pieces right at the load
// (every V element belongs to exactly one wave's fragment, so nothing is split twice).
// ABL: compile-time ablation mask for the diagnostic build (results are wrong, only timing matters; always 0 in
// the product): 1 weights loaded once, 2 no input transform, 4 no staging (loads + raw write), 8 no matrix work,
// 16 no epilogue.
template <int UPS, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_wino_bf16x3_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;
  float* V = lds + RAW_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = Winograd row xi of this wave's matrix work
  const int half = lane >> 5, l31 = lane & 31;

  int t = blockIdx.x;
  const int tx0 = t % p.tilesX;
  t /= p.tilesX;
  const int ty0 = t % p.tilesY;
  const int b = t / p.tilesY;
  const int nt = blockIdx.y;
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
      const int pix = (tid + i * 256) >> 2;
      const int pixc = pix < IN_PIX ? pix : IN_PIX - 1;
      const int yy = iy0 + pixc / IN_W, xx = ix0 + pixc % IN_W;
      const bool ok = pix < IN_PIX && yy >= 0 && yy < Hlim && xx >= 0 && xx < Wlim;
      const int yc = min(max(yy, 0), Hlim - 1), xc = min(max(xx, 0), Wlim - 1);
      const int sy = UPS ? (yc >> 1) : yc, sx = UPS ? (xc >> 1) : xc;
      poff[i] = (b * p.Hin + sy) * p.Win + sx;
      inside |= (ok ? 1u : 0u) << i;
    }
  }
  // this thread's two transform items (tile, channel quad cq, row xi): LDS offsets are chunk-invariant
  int rd_a[2], rd_b[2], wr_v[2];
  float sgn[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * 256;
    const int tile = item & 31, cq = (item >> 5) & 3, xi = item >> 7;
    const int ty = tile >> 3, tx = tile & 7;
    // rows of d that enter row xi of B^T d:  xi0: d0-d2, xi1: d1+d2, xi2: d2-d1, xi3: d1-d3
    const int ra_ = (xi == 0) ? 0 : (xi == 2 ? 2 : 1);
    const int rb_ = (xi == 0 || xi == 1) ? 2 : (xi == 2 ? 1 : 3);
    sgn[it] = (xi == 1) ? 1.f : -1.f;
    rd_a[it] = ((2 * ty + ra_) * IN_W + 2 * tx) * RAWP + cq * 4;
    rd_b[it] = ((2 * ty + rb_) * IN_W + 2 * tx) * RAWP + cq * 4;
    wr_v[it] = (((xi * 4) * 4 + cq) * NT + tile) * 4;  // position 4*xi + nu, k quarter cq
  }
  const int wr_raw0 = (tid >> 2) * RAWP + c4 * 4;  // slot i lives 64 pixels further

  floatx16 acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][nb][r] = 0.f;

  const int nchunks = p.nch0 + p.nch1;
  const int last = nchunks - 1;
  // packed U: [nt][chunk][pos 16][nb 2][piece 3][lane 64] x 16 B; this wave's positions are 4*wave .. 4*wave + 3
  const uint4* wu = reinterpret_cast<const uint4*>(p.wpack) + ((size_t)nt * nchunks * 16 + wave * 4) * POS_U4 + lane;
  const int nlin = nchunks * 4;  // (chunk, position-of-this-wave) pairs in matrix-phase order
  uint4 bq[2][6];
  auto load_b = [&](int buf, int lin) {
    const int l = lin < nlin ? lin : nlin - 1;
    const uint4* src = wu + ((size_t)(l >> 2) * 16 + (l & 3)) * POS_U4;
#pragma unroll
    for (int i = 0; i < 6; ++i) bq[buf][i] = src[i * 64];
  };
  // A fragment of position 4*wave + q: lane (tile l31, K half) reads k quarters 2*half, 2*half + 1
  const float* va = V + ((wave * 16 + 2 * half) * NT + l31) * 4;

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
  // registers (chunk ch) -> (prologue SiLU(a*x+b)) -> raw LDS tile
  auto raw_write = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    const bool pro = (p.in_coef != nullptr) && !s1;
    const int Csrc = s1 ? p.C1 : p.C0;
    const bool cvalid = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4 < Csrc;
    const unsigned m = cvalid ? inside : 0u;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if (i < NLOAD - 1 || ((tid + i * 256) >> 2) < IN_PIX) {
        float4 x = v[i];
        if (!((m >> i) & 1u)) {
          x = make_float4(0.f, 0.f, 0.f, 0.f);  // padding stays exactly zero: it pads the ACTIVATED tensor
        } else if (pro) {
          x.x = silu_fast(fmaf(ca.x, x.x, cb.x));
          x.y = silu_fast(fmaf(ca.y, x.y, cb.y));
          x.z = silu_fast(fmaf(ca.z, x.z, cb.z));
          x.w = silu_fast(fmaf(ca.w, x.w, cb.w));
        }
        st4(raw + wr_raw0 + i * (64 * RAWP), x);
      }
    }
  };
  // input transform V = B^T d B of the chunk in `raw` -> V[buf], two items per thread
  auto transform = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const float* pa = raw + rd_a[it];
      const float* pb = raw + rd_b[it];
      const float2v sg = float2v{sgn[it], sgn[it]};
      f4 w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f4 da = ldf4(pa + c * RAWP), db = ldf4(pb + c * RAWP);
        w[c].lo = da.lo + sg * db.lo;
        w[c].hi = da.hi + sg * db.hi;
      }
      float* vo = V + buf * V_FLOATS + wr_v[it];  // V[xi][0..3] = w0-w2, w1+w2, w2-w1, w1-w3
      stf4(vo, w[0] - w[2]);
      stf4(vo + 1 * (4 * NT * 4), w[1] + w[2]);
      stf4(vo + 2 * (4 * NT * 4), w[2] - w[1]);
      stf4(vo + 3 * (4 * NT * 4), w[1] - w[3]);
    }
  };

#define DMH_TERM(q, sa, sb)                                                                                       \
  acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa], __builtin_bit_cast(bf16x8, bq[(q) & 1][sb]), acc[q][0], \
                                                      0, 0, 0);                                                   \
  acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa], __builtin_bit_cast(bf16x8, bq[(q) & 1][3 + sb]),     \
                                                      acc[q][1], 0, 0, 0);
  // position 4*wave + q of chunk c: M[32 tiles][64 cout] += V[tiles][k] U[k][cout], smallest terms first
#define DMH_POSITION(c, q)                                                         \
  {                                                                                \
    if (!(ABL & 1)) load_b(((q) & 1) ^ 1, (c) * 4 + (q) + 1); /* next position's weights */ \
    __builtin_amdgcn_sched_barrier(0x38F);    /* memory loads stay up here */      \
    const float* vb = va + ((c) & 1) * V_FLOATS + (q) * (4 * NT * 4);              \
    const float4 f0 = ld4(vb), f1 = ld4(vb + NT * 4);                              \
    bf16x8 a[3];                                                                   \
    split8(f0, f1, a);                                                             \
    DMH_TERM(q, 2, 0)                                                              \
    DMH_TERM(q, 0, 2)                                                              \
    DMH_TERM(q, 1, 1)                                                              \
    DMH_TERM(q, 1, 0)                                                              \
    DMH_TERM(q, 0, 1)                                                              \
    DMH_TERM(q, 0, 0)                                                              \
  }

  auto stage = [&](int c, auto TR, auto RW) {
    if constexpr (decltype(TR)::value && !(ABL & 2)) transform((c & 1) ^ 1);
    if constexpr (!(ABL & 8)) {
      DMH_POSITION(c, 0)
      DMH_POSITION(c, 1)
    }
    __syncthreads();  // raw is free again (and V[(c+1)&1] is complete)
    if constexpr (decltype(RW)::value && !(ABL & 4)) {
      raw_write(c + 2);
      issue_chunk_loads(c + 3 < nchunks ? c + 3 : last);
    }
    if constexpr (!(ABL & 8)) {
      DMH_POSITION(c, 2)
      DMH_POSITION(c, 3)
    }
    __syncthreads();  // raw of chunk c+2 published; V[c & 1] is free
  };

  // ---- pipeline fill
  issue_chunk_loads(0);
  load_b(0, 0);
  if (ABL & 1) load_b(1, 1);
  raw_write(0);
  __syncthreads();
  issue_chunk_loads(1 < nchunks ? 1 : last);
  transform(0);
  __syncthreads();
  if (nchunks > 1) raw_write(1);
  issue_chunk_loads(2 < nchunks ? 2 : last);
  __syncthreads();

  for (int c = 0; c < nchunks - 2; ++c) stage(c, True_{}, True_{});
  if (nchunks >= 2) stage(nchunks - 2, True_{}, False_{});
  stage(nchunks - 1, False_{}, False_{});
#undef DMH_POSITION
#undef DMH_TERM

  // ---- output transform Y = A^T M A.  This wave holds row xi = wave of M: contract over nu in registers,
  //      Z[xi][j] = sum_nu A[nu][j] M[xi][nu], exchange rows through LDS, contract over xi while reading.
  float* Z = lds;  // (the last stage ended on a barrier: LDS is free)
  if constexpr (ABL & 16) {
    if (acc[0][0][0] + acc[1][1][3] + acc[2][0][5] + acc[3][1][7] == 12345.f) p.out[0] = 0.f;  // keep the matrix work alive
    return;
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * half;  // C/D layout of the 32x32 block: row = tile, col = cout
      const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r];
      float* z = Z + ((wave * 2) * NT + tile) * ZP + nb * 32 + l31;
      z[0] = m0 + m1 + m2;
      z[NT * ZP] = m1 - m2 - m3;
    }
  __syncthreads();
  EpilogueRows er(p, b, n0);
  {
    const int c4e = er.c4;
    const float* Zc = Z + c4e * 4;
    er.template store_rows_fn<TW>(
        p,
        [Zc, wave](int rr) {
          const int row = wave * 32 + rr;  // pixel (row >> 4, row & 15) of the 8x16 region
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
        wave * 32, oy0, ox0);
  }
  er.write_stats(p, lds, tile_in_sample);
}

// transformed weights U = G g G^T, split into three bf16 pieces, in fragment-major order
//   bf16 index = ((((((nt * nchunks + ch) * 16 + pos) * 2 + nb) * 3 + piece) * 64 + lane) * 8 + j
//   -> piece of U_pos[k = (lane >> 5) * 8 + j][cout = nt*64 + nb*32 + (lane & 31)]
__global__ void pack_winobx_weight_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int C0,
                                          int C1, int nch0, int nch1, int64_t total) {
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
    val = acc;
  }
  const float r1 = val - __uint_as_float(__float_as_uint(val) & 0xffff0000u);
  const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
  const float pc = piece == 0 ? val : (piece == 1 ? r1 : r2);
  wp[idx] = (unsigned short)(__float_as_uint(pc) >> 16);
}

int64_t dmh_winobx_pack_floats(int Cout, int C0, int C1) {
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * 16 * POS_U4 * 4;  // uint4 = 4 floats
}

int dmh_winobx_pack(const float* w, float* wpack, int Cout, int C0, int C1, hipStream_t st) {
  const int nch0 = cdiv(C0, KC), nch1 = cdiv(C1, KC);
  const int64_t total = dmh_winobx_pack_floats(Cout, C0, C1) * 2;  // bf16 elements
  hipLaunchKernelGGL(pack_winobx_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w,
                     reinterpret_cast<unsigned short*>(wpack), Cout, C0, C1, nch0, nch1, total);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight(winograd bf16x3)");
  return DMH_OK;
}

int dmh_winobx_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, TW);
  static bool attr = false;
  if (!attr) {
    hipError_t e0 = hipFuncSetAttribute((const void*)conv_wino_bf16x3_kernel<0>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipError_t e1 = hipFuncSetAttribute((const void*)conv_wino_bf16x3_kernel<1>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    DMH_REQUIRE(e0 == hipSuccess && e1 == hipSuccess, "dmh_conv2d: cannot raise the LDS limit");
    attr = true;
  }
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
#ifdef DMH_STAMPS
  if (const char* e = getenv("DMH_BX_ABL")) {
#define DMH_ABL_CASE(n)                                                                                          \
  case n:                                                                                                        \
    (void)hipFuncSetAttribute((const void*)conv_wino_bf16x3_kernel<0, n>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              LDS_BYTES);                                                                        \
    hipLaunchKernelGGL((conv_wino_bf16x3_kernel<0, n>), grid, dim3(256), LDS_BYTES, st, a);                      \
    return DMH_OK;
    switch (atoi(e)) {
      DMH_ABL_CASE(1)
      DMH_ABL_CASE(2)
      DMH_ABL_CASE(4)
      DMH_ABL_CASE(6)
      DMH_ABL_CASE(8)
      DMH_ABL_CASE(16)
      DMH_ABL_CASE(7)
      DMH_ABL_CASE(23)
      DMH_ABL_CASE(14)
      DMH_ABL_CASE(30)
    }
#undef DMH_ABL_CASE
  }
#endif
  if (d->upsample2)
    hipLaunchKernelGGL((conv_wino_bf16x3_kernel<1>), grid, dim3(256), LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL((conv_wino_bf16x3_kernel<0>), grid, dim3(256), LDS_BYTES, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d(winograd bf16x3)");
  return DMH_OK;
}
