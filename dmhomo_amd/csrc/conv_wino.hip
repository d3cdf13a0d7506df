// K1 fast path — 3x3 / stride-1 convolution by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        (Lavin & Gray; exact in real arithmetic)
//
// 16 multiplies per 2x2 outputs instead of 36: 2.25x fewer matrix FLOPs than the implicit GEMM, all in
// fp32 (the transforms only add, subtract and halve).  Per workgroup: an 8x16 output-pixel region
// (32 Winograd tiles) x 64 output channels; per 16-channel chunk
//   1. the 10x18 input halo is staged to LDS through the fused GroupNorm+SiLU prologue,
//   2. all 256 threads transform it to V[16 positions][k-quarter][32 tiles][4] (LDS, conflict-free b128),
//   3. each wave runs 16 positions x (16 tiles x 32 cout) on v_mfma_f32_16x16x4_f32, the transformed
//      weights U streaming from L2 in fragment-major order straight into the B operand registers.
// The accumulator layout of the 16x16 MFMA keeps all 16 positions of a (tile, cout) pair in ONE lane, so
// the output transform is register-local; results go through the shared float4 row epilogue
// (+bias, +residual, GroupNorm partials).  Dual-source (fused torch.cat) and nearest-x2 upsampled
// inputs are handled in the halo gather exactly as in conv.hip.
#include <stdlib.h>

#include "conv_args.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TH = 8, TW = 16, KC = 16, IN_H = 10, IN_W = 18, IN_PIX = IN_H * IN_W, RAWP = 20, NT = 32;
constexpr int RAW_FLOATS = IN_PIX * RAWP;          // 3600
constexpr int V_FLOATS = 16 * 4 * NT * 4;          // 8192: [pos][kq][tile][4]
constexpr int SLAB_FLOATS = 128 * EpilogueRows::EP;  // epilogue slab: 128 pixels x 68
constexpr int LDS_FLOATS = (RAW_FLOATS + V_FLOATS) > SLAB_FLOATS ? (RAW_FLOATS + V_FLOATS) : SLAB_FLOATS;
constexpr int PD = 4;                              // weight prefetch depth in positions
}  // namespace

typedef float float2v __attribute__((ext_vector_type(2)));
struct f4 {  // a float4 as two packed pairs: + and - compile to v_pk_add_f32 (two lane-ops per instruction)
  float2v lo, hi;
};
__device__ __forceinline__ f4 ldf4(const float* p) {
  const float4 t = ld4(p);
  f4 r;
  r.lo = float2v{t.x, t.y};
  r.hi = float2v{t.z, t.w};
  return r;
}
__device__ __forceinline__ void stf4(float* p, const f4& v) { st4(p, make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y)); }
__device__ __forceinline__ f4 operator+(const f4& a, const f4& b) { return f4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ f4 operator-(const f4& a, const f4& b) { return f4{a.lo - b.lo, a.hi - b.hi}; }

// NOTE on instruction economy: on gfx950 the fp32 MFMA runs at the fp32 VALU rate and does not overlap with
// VALU work of the same SIMD (measured: MFMA-busy + VALU-busy + LDS issue ~ 100 %), so every vector
// instruction in the chunk loop is paid in matrix throughput.  Everything chunk-invariant (halo pixel
// offsets, validity mask, LDS offsets of the transform items) is therefore computed once, global loads use
// a uniform (scalar) base + a precomputed 32-bit lane offset, and the transforms use packed fp32 adds.
// G = 1: 4 waves, 64 output channels per workgroup, two workgroups per CU.
// G = 2: 8 waves, 128 output channels per workgroup (one per CU): the staging + input transform of a chunk
//        is shared by twice the matrix work, for layers with Cout % 128 == 0.
template <int UPS, int G>
__global__ __launch_bounds__(256 * G, 2) void conv_wino_kernel(ConvArgs p) {
  constexpr int NTHR = 256 * G;
  constexpr int NLOAD = (IN_PIX * 4 + NTHR - 1) / NTHR;  // halo float4 slots per thread per chunk
  constexpr int NIT = 512 / NTHR;                        // transform items per thread per chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;
  float* V = lds + RAW_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gq = wave >> 2;                              // which 64-channel block of this workgroup
  const int wm = (wave >> 1) & 1, wn = wave & 1;         // 2 (tiles) x 2 (cout) waves per block
  const int j16 = lane & 15, kq = lane >> 4;

  int t = blockIdx.x;
  const int tx0 = t % p.tilesX;
  t /= p.tilesX;
  const int ty0 = t % p.tilesY;
  if (t / p.tilesY >= dmh_rows_n(p.rows, p.B)) return;  // (DmhConv.rows: the workgroups of inactive rows retire)
  const int b = dmh_rows_phys(p.rows, t / p.tilesY);
  const int nt = blockIdx.y * G + gq;
  const int n0 = nt * 64;
  const int tile_in_sample = ty0 * p.tilesX + tx0;
  const int oy0 = ty0 * TH, ox0 = tx0 * TW;

#ifdef DMH_STAMPS
  // diagnostic build only (make stamps): per-wave cycle totals of each phase, written over this tile's stats slot
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
#else
#define STAMP(i)
#endif

  // ---- chunk-invariant staging state: 3 halo slots per thread (pixel, channel quad c4)
  const int c4 = tid & 3;
  int poff[NLOAD];      // clamped source pixel index (within the whole tensor), always addressable
  unsigned inside = 0;  // bit i: slot i is image data (not zero padding / not beyond the tile)
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
  // LDS offsets of this thread's two transform items (tile, channel quad cq, row xi): chunk-invariant too
  int rd_a[NIT], rd_b[NIT], wr_v[NIT];
  float sgn[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + it * NTHR;
    const int tile = item & 31, cq = (item >> 5) & 3, xi = item >> 7;
    const int ty = tile >> 3, tx = tile & 7;
    // rows of d that enter row xi of B^T d:  xi0: d0-d2, xi1: d1+d2, xi2: d2-d1, xi3: d1-d3
    const int ra_ = (xi == 0) ? 0 : (xi == 2 ? 2 : 1);
    const int rb_ = (xi == 0 || xi == 1) ? 2 : (xi == 2 ? 1 : 3);
    sgn[it] = (xi == 1) ? 1.f : -1.f;
    rd_a[it] = ((2 * ty + ra_) * IN_W + 2 * tx) * RAWP + cq * 4;
    rd_b[it] = ((2 * ty + rb_) * IN_W + 2 * tx) * RAWP + cq * 4;
    wr_v[it] = (((xi * 4) * 4 + cq) * NT + tile) * 4;  // position 4*xi + nu, k-quarter cq
  }
  const int wr_raw0 = (tid >> 2) * RAWP + c4 * 4;  // slot i lives NTHR/4 pixels further

  floatx4 acc[16][2];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][nb][r] = 0.f;

  const int nchunks = p.nch0 + p.nch1;
  // packed U: [nt][chunk][pos 16][nb4 4][lane 64][4]; this wave's two 16-channel blocks are nb4 = 2*wn, 2*wn+1.
  // uniform (scalar) base + lane offset => saddr addressing, pointer bumps on the scalar unit
  const float* wu = p.wpack + (size_t)nt * nchunks * (16 * 4 * 256) + (wn * 2) * 256;
  const int wlane = lane * 4;
  const float* va = V + (kq * NT + wm * 16 + j16) * 4;

  float4 v[NLOAD];
  float4 ca = make_float4(1.f, 1.f, 1.f, 1.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue_chunk_loads = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    const float* src = s1 ? p.src1 : p.src0;  // uniform
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

  float4 bq[PD][2];
  issue_chunk_loads(0);

  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- 1. registers -> (prologue SiLU(a*x+b)) -> raw LDS tile
#ifdef DMH_STAMPS
    if (!(p.ablate & 1))
#endif
    {
      const bool s1 = ch >= p.nch0;
      const bool pro = (p.in_coef != nullptr) && !s1;
      const int Csrc = s1 ? p.C1 : p.C0;
      const bool cvalid = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4 < Csrc;
      const unsigned m = cvalid ? inside : 0u;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        if (i < NLOAD - 1 || ((tid + i * NTHR) >> 2) < IN_PIX) {
          float4 x = v[i];
          if (!((m >> i) & 1u)) {
            x = make_float4(0.f, 0.f, 0.f, 0.f);  // padding stays exactly zero: it pads the ACTIVATED tensor
          } else if (pro) {
            x.x = silu_fast(fmaf(ca.x, x.x, cb.x));
            x.y = silu_fast(fmaf(ca.y, x.y, cb.y));
            x.z = silu_fast(fmaf(ca.z, x.z, cb.z));
            x.w = silu_fast(fmaf(ca.w, x.w, cb.w));
          }
          st4(raw + wr_raw0 + i * ((NTHR / 4) * RAWP), x);
        }
      }
    }
    // first PD weight positions of THIS chunk: issued only now, so that the vmcnt wait of the raw write above
    // covered nothing but the halo loads (refills never cross a chunk boundary)
    const float* wch = wu + (size_t)ch * (16 * 4 * 256);
#pragma unroll
    for (int q = 0; q < PD; ++q) {
      bq[q][0] = ld4(wch + q * 1024 + wlane);
      bq[q][1] = ld4(wch + q * 1024 + 256 + wlane);
    }
    STAMP(0)          // raw write (+ prologue)
    __syncthreads();  // raw published; every wave has also left the previous matrix phase (V is free)
    STAMP(1)          // barrier 1
    // next chunk's halo (on the last chunk: a harmless re-load of itself — an UNCONDITIONAL issue lets hipcc
    // count vmcnt exactly instead of draining these loads at the first weight wait of the matrix phase)
    issue_chunk_loads(ch + 1 < nchunks ? ch + 1 : ch);

    // ---- 2. input transform V = B^T d B, two items per thread, packed fp32 adds
#ifdef DMH_STAMPS
    if (!(p.ablate & 1))
#endif
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
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
      float* vo = V + wr_v[it];  // V[xi][0..3] = w0-w2, w1+w2, w2-w1, w1-w3
      stf4(vo, w[0] - w[2]);
      stf4(vo + 1 * (4 * NT * 4), w[1] + w[2]);
      stf4(vo + 2 * (4 * NT * 4), w[2] - w[1]);
      stf4(vo + 3 * (4 * NT * 4), w[1] - w[3]);
    }
    STAMP(2)          // input transform
    __syncthreads();  // V published
    STAMP(3)          // barrier 2

    // ---- 3. matrix phase: M_pos[tile][cout] += V_pos[tile][k] * U_pos[k][cout], 16 positions
#ifdef DMH_STAMPS
    if (!(p.ablate & 2))
#endif
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 a = ld4(va + q * (4 * NT * 4));
      const float4 b0 = bq[q % PD][0], b1 = bq[q % PD][1];
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[q][1], 0, 0, 0);
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[q][1], 0, 0, 0);
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc[q][1], 0, 0, 0);
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc[q][1], 0, 0, 0);
      if (q + PD < 16) {  // refill this slot with the position PD ahead
        bq[q % PD][0] = ld4(wch + (q + PD) * 1024 + wlane);
        bq[q % PD][1] = ld4(wch + (q + PD) * 1024 + 256 + wlane);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the refill load here, not next to its consumer
    }
    STAMP(4)  // matrix phase
  }

  // ---- output transform Y = A^T M A, register-local: lane holds M_pos[tile = wm*16 + kq*4 + r][cout = .. + j16]
  __syncthreads();  // all waves left the last matrix phase: LDS becomes the pixel x channel slab
  constexpr int EP = EpilogueRows::EP;
  float* slab = lds + gq * SLAB_FLOATS;  // one 128-pixel x 64-channel slab per 4-wave group
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int col = (wn * 2 + nb) * 16 + j16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tile = wm * 16 + kq * 4 + r;
      const int ty = tile >> 3, tx = tile & 7;
      float tt[2][4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        const float m0 = acc[0 + nu][nb][r], m1 = acc[4 + nu][nb][r], m2 = acc[8 + nu][nb][r], m3 = acc[12 + nu][nb][r];
        tt[0][nu] = m0 + m1 + m2;
        tt[1][nu] = m1 - m2 - m3;
      }
      float* sl = slab + ((2 * ty) * TW + 2 * tx) * EP + col;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        sl[(i * TW) * EP] = tt[i][0] + tt[i][1] + tt[i][2];
        sl[(i * TW + 1) * EP] = tt[i][1] - tt[i][2] - tt[i][3];
      }
    }
  }
  __syncthreads();
  EpilogueRows er(p, b, n0);
  er.template store_rows<TW>(p, slab + (wave & 3) * (32 * EP), (wave & 3) * 32, oy0, ox0);
#ifdef DMH_STAMPS
  STAMP(5)  // output transform + row epilogue
  if (p.stats && lane == 0 && nt == 0 && G == 1) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(
                                p.stats + ((size_t)(b * p.tilesX * p.tilesY + tile_in_sample) * p.Cout) * 2) + wave * 8;
    for (int i = 0; i < 6; ++i) d[i] = tk[i];
    d[6] = t_now - t_begin;
    d[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11)) << 32) |
           __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_REG_LDS_ALLOC : HW_REG_HW_ID
  }
#else
  er.write_stats(p, lds, tile_in_sample);
#endif
}

// transformed weights U = G g G^T in fragment-major order [nt][chunk][pos][nb4][lane][4]:
// lane = kq*16 + j holds k = 4*kq + e (e = 0..3) of output channel nt*64 + nb4*16 + j
__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C0, int C1,
                                        int nch0, int nch1, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int64_t r = idx;
  const int e = r % 4;
  r /= 4;
  const int lane = r % 64;
  r /= 64;
  const int nb4 = r % 4;
  r /= 4;
  const int pos = r % 16;
  r /= 16;
  const int ch = r % (nch0 + nch1);
  const int nt = r / (nch0 + nch1);
  const int o = nt * 64 + nb4 * 16 + (lane & 15);
  const int k = 4 * (lane >> 4) + e;
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
  wp[idx] = val;
}

int64_t dmh_wino_pack_floats(int Cout, int C0, int C1) {
  return (int64_t)cdiv(Cout, 64) * (cdiv(C0, KC) + cdiv(C1, KC)) * 16 * 4 * 256;
}

int dmh_wino_pack(const float* w, float* wpack, int Cout, int C0, int C1, hipStream_t st) {
  const int nch0 = cdiv(C0, KC), nch1 = cdiv(C1, KC);
  const int64_t total = dmh_wino_pack_floats(Cout, C0, C1);
  hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, w, wpack, Cout, C0,
                     C1, nch0, nch1, total);
  DMH_CHECK_LAUNCH("dmh_pack_conv_weight(winograd)");
  return DMH_OK;
}

int dmh_wino_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, TW);
#ifdef DMH_STAMPS
  {
    const char* e = getenv("DMH_WINO_ABLATE");
    a.ablate = e ? atoi(e) : 0;
  }
#endif
  static int wide = -1;  // development knob: DMH_WINO_WIDE=0 disables the 128-channel workgroups
  if (wide < 0) {
    const char* e = getenv("DMH_WINO_WIDE");
    wide = e ? atoi(e) : 1;
  }
  if (wide && a.Cout % 128 == 0) {
    constexpr int LDSW = (2 * SLAB_FLOATS > LDS_FLOATS ? 2 * SLAB_FLOATS : LDS_FLOATS) * 4;  // 69.6 KB
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv_wino_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSW);
      (void)hipFuncSetAttribute((const void*)conv_wino_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSW);
      attr = true;
    }
    dim3 grid(a.tilesX * a.tilesY * a.B, a.Cout / 128);
    if (d->upsample2)
      hipLaunchKernelGGL((conv_wino_kernel<1, 2>), grid, dim3(512), LDSW, st, a);
    else
      hipLaunchKernelGGL((conv_wino_kernel<0, 2>), grid, dim3(512), LDSW, st, a);
    DMH_CHECK_LAUNCH("dmh_conv2d(winograd, 128-channel workgroups)");
    return DMH_OK;
  }
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
  if (d->upsample2)
    hipLaunchKernelGGL((conv_wino_kernel<1, 1>), grid, dim3(256), LDS_FLOATS * 4, st, a);
  else
    hipLaunchKernelGGL((conv_wino_kernel<0, 1>), grid, dim3(256), LDS_FLOATS * 4, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d(winograd)");
  return DMH_OK;
}
