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
#include "conv_args.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int TH = 8, TW = 16, KC = 16, IN_H = 10, IN_W = 18, IN_PIX = IN_H * IN_W, RAWP = 20, NT = 32;
constexpr int RAW_FLOATS = IN_PIX * RAWP;          // 3600
constexpr int V_FLOATS = 16 * 4 * NT * 4;          // 8192: [pos][kq][tile][4]
constexpr int SLAB_FLOATS = 128 * EpilogueRows::EP;  // epilogue slab: 128 pixels x 68
constexpr int LDS_FLOATS = (RAW_FLOATS + V_FLOATS) > SLAB_FLOATS ? (RAW_FLOATS + V_FLOATS) : SLAB_FLOATS;
constexpr int NLOAD = (IN_PIX * 4 + 255) / 256;    // 3 float4 per thread per chunk
constexpr int PD = 4;                              // weight prefetch depth in positions
}  // namespace

template <int UPS>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;
  float* V = lds + RAW_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2 (tiles) x 2 (cout) waves
  const int j16 = lane & 15, kq = lane >> 4;

  int t = blockIdx.x;
  const int tx0 = t % p.tilesX;
  t /= p.tilesX;
  const int ty0 = t % p.tilesY;
  const int b = t / p.tilesY;
  const int nt = blockIdx.y;
  const int n0 = nt * 64;
  const int tile_in_sample = ty0 * p.tilesX + tx0;
  const int oy0 = ty0 * TH, ox0 = tx0 * TW;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int Hlim = UPS ? p.Hin * 2 : p.Hin;
  const int Wlim = UPS ? p.Win * 2 : p.Win;

  floatx4 acc[16][2];
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][nb][r] = 0.f;

  const int nchunks = p.nch0 + p.nch1;
  // packed U: [nt][chunk][pos 16][nb4 4][lane 64][4]; this wave's two 16-channel blocks are nb4 = 2*wn, 2*wn+1
  const float* wbase = p.wpack + (size_t)nt * nchunks * (16 * 4 * 256) + (wn * 2) * 256 + lane * 4;

  // ---- input halo: global -> registers
  const int c4 = tid & 3;
  float4 v[NLOAD];
  float4 ca, cb;
  unsigned inside;
  auto issue_chunk_loads = [&](int ch) {
    const bool s1 = ch >= p.nch0;
    const float* src = s1 ? p.src1 : p.src0;
    const int Csrc = s1 ? p.C1 : p.C0;
    const int c = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4;
    const bool cvalid = c < Csrc;
    inside = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int pix = (tid + i * 256) >> 2;
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
    ca = make_float4(1.f, 1.f, 1.f, 1.f);
    cb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.in_coef != nullptr && !s1 && cvalid) {
      ca = ld4(p.in_coef + (size_t)(b * 2 + 0) * p.C0 + c);
      cb = ld4(p.in_coef + (size_t)(b * 2 + 1) * p.C0 + c);
    }
  };

  // ---- weight (B operand) pipeline: PD positions in flight, refilled in place right after use
  float4 bq[PD][2];
  auto load_b = [&](int slot, int ch, int pos) {
    const float* wsrc = wbase + ((size_t)ch * 16 + pos) * (4 * 256);
    bq[slot][0] = ld4(wsrc);
    bq[slot][1] = ld4(wsrc + 256);
  };
#pragma unroll
  for (int q = 0; q < PD; ++q) load_b(q, 0, q);
  issue_chunk_loads(0);

  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- 1. registers -> (prologue SiLU(a*x+b)) -> raw LDS tile.  The previous chunk's transform finished
    //         reading `raw` before the barrier that preceded its matrix phase.
    {
      const bool pro = (p.in_coef != nullptr) && ch < p.nch0;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int pix = (tid + i * 256) >> 2;
        if (pix < IN_PIX) {
          float4 x = v[i];
          if (pro && ((inside >> i) & 1u)) {  // padding stays exactly zero: it pads the ACTIVATED tensor
            x.x = silu_fast(fmaf(ca.x, x.x, cb.x));
            x.y = silu_fast(fmaf(ca.y, x.y, cb.y));
            x.z = silu_fast(fmaf(ca.z, x.z, cb.z));
            x.w = silu_fast(fmaf(ca.w, x.w, cb.w));
          }
          st4(raw + pix * RAWP + c4 * 4, x);
        }
      }
    }
    __syncthreads();  // raw published; every wave has also left the previous matrix phase (V is free)
    if (ch + 1 < nchunks) issue_chunk_loads(ch + 1);  // travels behind the transform + matrix phase

    // ---- 2. input transform V = B^T d B: work item = (tile, channel quad, row xi), two per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + it * 256;
      const int tile = item & 31, cq = (item >> 5) & 3, xi = item >> 7;
      const int ty = tile >> 3, tx = tile & 7;
      // rows of d that enter row xi of B^T d:  xi0: d0-d2, xi1: d1+d2, xi2: d2-d1, xi3: d1-d3
      const int ra_ = (xi == 0) ? 0 : (xi == 2 ? 2 : 1);
      const int rb_ = (xi == 0 || xi == 1) ? 2 : (xi == 2 ? 1 : 3);
      const float sgn = (xi == 1) ? 1.f : -1.f;
      const float* pa = raw + ((2 * ty + ra_) * IN_W + 2 * tx) * RAWP + cq * 4;
      const float* pb = raw + ((2 * ty + rb_) * IN_W + 2 * tx) * RAWP + cq * 4;
      float4 w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 da = ld4(pa + c * RAWP), db = ld4(pb + c * RAWP);
        w[c].x = fmaf(sgn, db.x, da.x);
        w[c].y = fmaf(sgn, db.y, da.y);
        w[c].z = fmaf(sgn, db.z, da.z);
        w[c].w = fmaf(sgn, db.w, da.w);
      }
      float4 o0, o1, o2, o3;  // V[xi][0..3] = w0-w2, w1+w2, w2-w1, w1-w3
      o0.x = w[0].x - w[2].x; o0.y = w[0].y - w[2].y; o0.z = w[0].z - w[2].z; o0.w = w[0].w - w[2].w;
      o1.x = w[1].x + w[2].x; o1.y = w[1].y + w[2].y; o1.z = w[1].z + w[2].z; o1.w = w[1].w + w[2].w;
      o2.x = w[2].x - w[1].x; o2.y = w[2].y - w[1].y; o2.z = w[2].z - w[1].z; o2.w = w[2].w - w[1].w;
      o3.x = w[1].x - w[3].x; o3.y = w[1].y - w[3].y; o3.z = w[1].z - w[3].z; o3.w = w[1].w - w[3].w;
      float* vo = V + (((xi * 4) * 4 + cq) * NT + tile) * 4;  // position 4*xi + nu, k-quarter cq
      st4(vo, o0);
      st4(vo + 1 * (4 * NT * 4), o1);
      st4(vo + 2 * (4 * NT * 4), o2);
      st4(vo + 3 * (4 * NT * 4), o3);
    }
    __syncthreads();  // V published

    // ---- 3. matrix phase: M_pos[tile][cout] += V_pos[tile][k] * U_pos[k][cout], 16 positions
    const float* va = V + (kq * NT + wm * 16 + j16) * 4;
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
      // refill this slot with the position PD ahead (wraps into the next chunk)
      {
        int nq = q + PD, nch = ch;
        if (nq >= 16) {
          nq -= 16;
          nch = ch + 1;
        }
        if (nch < nchunks) load_b(q % PD, nch, nq);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the refill load here, not next to its consumer
    }
  }

  // ---- output transform Y = A^T M A, register-local: lane holds M_pos[tile = wm*16 + kq*4 + r][cout = .. + j16]
  __syncthreads();  // all waves left the last matrix phase: LDS becomes the pixel x channel slab
  constexpr int EP = EpilogueRows::EP;
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
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int prow = (2 * ty + i) * TW + 2 * tx;
        lds[prow * EP + col] = tt[i][0] + tt[i][1] + tt[i][2];
        lds[(prow + 1) * EP + col] = tt[i][1] - tt[i][2] - tt[i][3];
      }
    }
  }
  __syncthreads();
  EpilogueRows er(p, b, n0);
  er.template store_rows<TW>(p, lds + wave * (32 * EP), wave * 32, oy0, ox0);
  er.write_stats(p, lds, tile_in_sample);
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
  dim3 grid(a.tilesX * a.tilesY * a.B, cdiv(a.Cout, 64));
  if (d->upsample2)
    hipLaunchKernelGGL((conv_wino_kernel<1>), grid, dim3(256), LDS_FLOATS * 4, st, a);
  else
    hipLaunchKernelGGL((conv_wino_kernel<0>), grid, dim3(256), LDS_FLOATS * 4, st, a);
  DMH_CHECK_LAUNCH("dmh_conv2d(winograd)");
  return DMH_OK;
}
