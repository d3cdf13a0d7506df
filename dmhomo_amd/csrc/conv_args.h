// Shared pieces of the convolution kernels: launch arguments and the float4 row epilogue.
#pragma once
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
  const float* oscale;  // per-output-channel power-of-two factor undoing the weight scaling (conv_f16x3.hip), or null
  const float* in_bound;  // null or [B][in_bound_n]: upper bounds of the prologue's |a*x + b| per sample (DmhConv.in_bound)
  int in_bound_n;
  int fin_n;             // DmhConv.fin_*: the pointwise projection of the finished pixel (0: none)
  const float* fin_w;
  const float* fin_b;
  float* fin_out;
  float* pix_stats;      // DmhConv.pix_stats / pix_eps (null: none)
  float pix_eps;
  int B, Hin, Win, C0, C1, Cout, Hout, Wout;
  int nch0, nch1, tilesX, tilesY;
  int ablate;  // diagnostic builds only (DMH_STAMPS): bit 0 skips staging + transform, bit 1 skips the matrix phase
  int xcd;     // 1: workgroup ids are re-dealt so that each XCD (id % 8) walks a contiguous run of tiles (DMH_CONV_XCD)
  const int32_t* rows;  // DmhConv.rows: null, or the active row subset of this launch (common.h: dmh_rows_n / dmh_rows_phys)
};


static inline ConvArgs fill_conv_args(const DmhConv* d, int Hout, int Wout, int KC, int TH, int TW) {
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
  a.oscale = nullptr;
  a.in_bound = d->in_coef ? d->in_bound : nullptr;
  a.in_bound_n = d->in_bound_n;
  a.fin_n = d->fin_w ? d->fin_n : 0;
  a.fin_w = d->fin_w;
  a.fin_b = d->fin_b;
  a.fin_out = d->fin_out;
  a.pix_stats = d->pix_stats;
  a.pix_eps = d->pix_eps;
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
  a.ablate = 0;
  static const int xcd = [] {
    const char* e = getenv("DMH_CONV_XCD");
    return e ? atoi(e) : 1;
  }();
  a.xcd = xcd;
  a.rows = d->rows;
  return a;
}


// Winograd 3x3 path (conv_wino.hip)
int64_t dmh_wino_pack_floats(int Cout, int C0, int C1);
int dmh_wino_pack(const float* w, float* wpack, int Cout, int C0, int C1, hipStream_t st);
int dmh_wino_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st);


// fp16 split path: two activation pieces x three weight planes, block-scaled (conv_f16x3.hip)
int64_t dmh_f16x3_pack_floats(int Cout, int C0, int C1, int KH, int KW);
int dmh_f16x3_pack(const float* w, float* wpack, int Cout, int C0, int C1, int KH, int KW, hipStream_t st);
int dmh_f16x3_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st);
int dmh_f16x3_pack_multi(const DmhPackJob* jobs, int njobs, float eps, hipStream_t st);
int64_t dmh_f16x3_up2_pack_floats(int Cout, int C0);
int dmh_f16x3_up2_pack(const float* w, float* wpack, int Cout, int C0, hipStream_t st);
int dmh_f16x3_launch_up2(const DmhConv* d, int Hout, int Wout, hipStream_t st);


// Second half of every conv epilogue: an LDS slab holds rows = output pixels x 64 channels (pitch EP);
// each wave turns 32 slab rows into NHWC float4 stores: + bias, + residual (optionally through
// SiLU(a*res+b)), and accumulates the per-channel (sum, sum^2) that GroupNorm needs.  Lane l handles
// channels 4*(l&15)..+3 of rows (l>>4) + 4*i.  Deterministic: fixed order, no atomics.
struct EpilogueRows {
  static constexpr int EP = 68;  // slab pitch in floats (64 + 4)
  int b, n0, chn, c4, rsub;
  bool cok;
  float4 bias, ra, rb, s1, s2, osc;

  __device__ __forceinline__ EpilogueRows(const ConvArgs& p, int b_, int n0_) : b(b_), n0(n0_) {
    const int lane = threadIdx.x & 63;
    c4 = lane & 15;
    rsub = lane >> 4;
    chn = n0 + c4 * 4;
    cok = chn < p.Cout;  // Cout % 4 == 0
    bias = make_float4(0.f, 0.f, 0.f, 0.f);
    ra = bias;
    rb = bias;
    s1 = bias;
    s2 = bias;
    osc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (cok && p.oscale) osc = ld4(p.oscale + chn);
    if (cok && p.bias) bias = ld4(p.bias + chn);
    if (cok && p.res_coef) {
      ra = ld4(p.res_coef + (size_t)(b * 2 + 0) * p.Cout + chn);
      rb = ld4(p.res_coef + (size_t)(b * 2 + 1) * p.Cout + chn);
    }
  }

  // wl: this wave's 32 slab rows; row_base: tile-row index of slab row 0 (row -> pixel (row / TW, row % TW))
  // (mul, dy, dx): the tile's pixel (y, x) lands on output pixel (mul*y + dy, mul*x + dx) — (2, parity) for the sub-pixel
  // form of Upsample + conv3x3, whose tiles live on the low-resolution grid
  template <int TW, bool FIN = false>
  __device__ __forceinline__ void store_rows(const ConvArgs& p, const float* wl, int row_base, int oy0, int ox0,
                                             int mul = 1, int dy = 0, int dx = 0) {
    const int c4_ = c4;
    store_rows_fn<TW, 8, FIN>(p, [wl, c4_](int rr) { return ld4(wl + rr * EP + c4_ * 4); }, row_base, oy0, ox0, mul, dy, dx);
  }

  // same, the 32 rows x this lane's channel quad coming from fetch(rr) instead of a plain slab
  // The residual rows of all 8 row groups of a 32-row pass are requested FIRST, unconditionally and from clamped addresses,
  // so that they are in flight together: loaded one by one inside the guarded loop (round 1) each of them exposed a full
  // memory round trip; the 128->64 @128^2 res_conv took 240 us with this epilogue against 140 us without a residual
  // (round 2: 190 us in isolation; in the two-stream step the stall was already covered by the other stream's kernels —
  // no change of images/s.  Requesting the rows ahead of the LDS transposition as well cost the launches WITHOUT a
  // residual 4 % and gained nothing: not kept).
  size_t oo[8];
  bool okr[8];
  float4 rv[8];
  bool pre = false;
  template <int TW, int NR = 8>
  __device__ __forceinline__ void prefetch_rows(const ConvArgs& p, int row_base, int oy0, int ox0, int mul = 1, int dy = 0,
                                                int dx = 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int row = row_base + i * 4 + rsub;
      const int oy = (oy0 + row / TW) * mul + dy, ox = (ox0 + row % TW) * mul + dx;
      okr[i] = cok && oy < p.Hout && ox < p.Wout;
      const int oyc = min(oy, p.Hout - 1), oxc = min(ox, p.Wout - 1);
      oo[i] = ((size_t)(b * p.Hout + oyc) * p.Wout + oxc) * p.Cout + (cok ? chn : 0);
    }
    if (p.res) {
#pragma unroll
      for (int i = 0; i < NR; ++i) rv[i] = ld4(p.res + oo[i]);
    }
    pre = true;
  }

  template <int TW, int NR = 8, bool FIN = false, typename Fetch>
  __device__ __forceinline__ void store_rows_fn(const ConvArgs& p, Fetch fetch, int row_base, int oy0, int ox0,
                                                int mul = 1, int dy = 0, int dx = 0) {
    if (!pre) prefetch_rows<TW, NR>(p, row_base, oy0, ox0, mul, dy, dx);
    pre = false;
    // all 8 rows are fetched and finished on every lane; only the store and the GroupNorm partials look at the row's
    // validity (round 1 wrapped each row in its own exec-masked region, which pinned every slab read behind the previous
    // row's store)
    float4 val[8];
#pragma unroll
    for (int i = 0; i < NR; ++i) val[i] = fetch(i * 4 + rsub);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      val[i].x = fmaf(val[i].x, osc.x, bias.x);  // osc == 1 unless the kernel scaled its weights: then exactly val + bias
      val[i].y = fmaf(val[i].y, osc.y, bias.y);
      val[i].z = fmaf(val[i].z, osc.z, bias.z);
      val[i].w = fmaf(val[i].w, osc.w, bias.w);
    }
    if (p.res) {
      if (p.res_coef) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          val[i].x += silu_fast(fmaf(ra.x, rv[i].x, rb.x));
          val[i].y += silu_fast(fmaf(ra.y, rv[i].y, rb.y));
          val[i].z += silu_fast(fmaf(ra.z, rv[i].z, rb.z));
          val[i].w += silu_fast(fmaf(ra.w, rv[i].w, rb.w));
        }
      } else {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          val[i].x += rv[i].x;
          val[i].y += rv[i].y;
          val[i].z += rv[i].z;
          val[i].w += rv[i].w;
        }
      }
    }
    if constexpr (FIN) {
      // DmhConv.fin_*: the 16 lanes of a pixel hold its (<= 64) channels as quads — exactly final_conv_kernel's layout, and
      // its arithmetic: per lane an fma chain over the quad from 0, row16_sum, + bias (bitwise the separate launch)
      if (p.fin_n > 0) {
        const int hw = p.Hout * p.Wout;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          if (o < p.fin_n) {
            const float4 wq = cok ? ld4(p.fin_w + (size_t)o * p.Cout + chn) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bo = p.fin_b ? p.fin_b[o] : 0.f;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
              float acc = fmaf(val[i].x, wq.x, 0.f);
              acc = fmaf(val[i].y, wq.y, acc);
              acc = fmaf(val[i].z, wq.z, acc);
              acc = fmaf(val[i].w, wq.w, acc);
              const float t = row16_sum(acc);
              const int row = row_base + i * 4 + rsub;
              const int oy = (oy0 + row / TW) * mul + dy, ox = (ox0 + row % TW) * mul + dx;
              if (c4 == o && oy < p.Hout && ox < p.Wout) p.fin_out[((size_t)b * p.fin_n + o) * hw + oy * p.Wout + ox] = t + bo;
            }
          }
        }
      }
      // DmhConv.pix_stats (Cout == 64: the 16 lanes of a pixel hold all of its channels): pixel_stats_kernel<16, 1>'s two
      // passes on the registers — sum, mean, centred second moment, in its order of operations
      if (p.pix_stats) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          float s = (val[i].x + val[i].y) + (val[i].z + val[i].w);
          s = lanes_sum<16>(s);
          const float mean = s / 64.0f;
          const float dx_ = val[i].x - mean, dy_ = val[i].y - mean, dz_ = val[i].z - mean, dw_ = val[i].w - mean;
          float qsum = (dx_ * dx_ + dy_ * dy_) + (dz_ * dz_ + dw_ * dw_);
          qsum = lanes_sum<16>(qsum);
          if (c4 == 0 && okr[i]) {
            float* ps = p.pix_stats + (oo[i] / (size_t)p.Cout) * 2;
            ps[0] = mean;
            ps[1] = 1.0f / sqrtf(qsum / 64.0f + p.pix_eps);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (okr[i] && (!FIN || p.out)) st4(p.out + oo[i], val[i]);
      const float k = okr[i] ? 1.f : 0.f;     // rows outside the image do not enter the statistics
      const float vx = val[i].x * k, vy = val[i].y * k, vz = val[i].z * k, vw = val[i].w * k;
      s1.x += vx;
      s1.y += vy;
      s1.z += vz;
      s1.w += vw;
      s2.x = fmaf(vx, vx, s2.x);
      s2.y = fmaf(vy, vy, s2.y);
      s2.z = fmaf(vz, vz, s2.z);
      s2.w = fmaf(vw, vw, s2.w);
    }
  }

  // Variant for the WM x WN wave grids of conv_f16x3.hip (each wave: 64 pixels x its own 64-channel block).
  // Stat tiles are 8 rows x 16 columns (= two waves along M) whatever the workgroup tile, so the tile count
  // does not depend on Cout: stats[b][stat tile][Cout][2] with stat tile = (ty * TH/8 + g) * tilesX + tx.
  template <int WM, int WN, int TH>
  __device__ __forceinline__ void write_stats_grid(const ConvArgs& p, float* lds, int ty, int tx, int by = -1) {
    if (by < 0) by = blockIdx.y;  // (kernels that re-deal their workgroup ids pass the logical cout-tile index)
    if (!p.stats) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    s1.x = rows_sum(s1.x);
    s1.y = rows_sum(s1.y);
    s1.z = rows_sum(s1.z);
    s1.w = rows_sum(s1.w);
    s2.x = rows_sum(s2.x);
    s2.y = rows_sum(s2.y);
    s2.z = rows_sum(s2.z);
    s2.w = rows_sum(s2.w);
    __syncthreads();
    float* red = lds;
    if (lane < 16) {
      float* q = red + (wave * 64 + c4 * 4) * 2;
      q[0] = s1.x;
      q[1] = s2.x;
      q[2] = s1.y;
      q[3] = s2.y;
      q[4] = s1.z;
      q[5] = s2.z;
      q[6] = s1.w;
      q[7] = s2.w;
    }
    __syncthreads();
    if (tid < 128) {
      const int q = tid >> 6, c = tid & 63;
      const int g = WN == 1 ? q : 0, wnq = WN == 1 ? 0 : q;
      const int chan = (by * WN + wnq) * 64 + c;
      const int srow = ty * (TH / 8) + g;
      if (chan < p.Cout && srow * 8 < p.Hout) {
        const int w0 = (g * 2 + 0) * WN + wnq, w1 = (g * 2 + 1) * WN + wnq;
        const float a0 = red[(w0 * 64 + c) * 2 + 0] + red[(w1 * 64 + c) * 2 + 0];
        const float a1 = red[(w0 * 64 + c) * 2 + 1] + red[(w1 * 64 + c) * 2 + 1];
        const int stiles = ((p.Hout + 7) / 8) * p.tilesX;
        float* st = p.stats + ((size_t)(b * stiles + srow * p.tilesX + tx) * p.Cout + chan) * 2;
        st[0] = a0;
        st[1] = a1;
      }
    }
  }

  // cross-lane + cross-wave reduction of the partials -> stats[b][tile][Cout][2].  Every group of 4 waves
  // (256 threads) owns one 64-channel block (n0 already points at it) and reduces through its own LDS patch.
  __device__ __forceinline__ void write_stats(const ConvArgs& p, float* lds, int tile_in_sample) {
    if (!p.stats) return;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, tid = threadIdx.x & 255;
    lds += (threadIdx.x >> 8) * 512;
    s1.x = rows_sum(s1.x);
    s1.y = rows_sum(s1.y);
    s1.z = rows_sum(s1.z);
    s1.w = rows_sum(s1.w);
    s2.x = rows_sum(s2.x);
    s2.y = rows_sum(s2.y);
    s2.z = rows_sum(s2.z);
    s2.w = rows_sum(s2.w);
    __syncthreads();  // every wave is done with the slab: reuse LDS as the cross-wave scratch
    float* red = lds;
    if (lane < 16) {
      float* q = red + (wave * 64 + c4 * 4) * 2;
      q[0] = s1.x;
      q[1] = s2.x;
      q[2] = s1.y;
      q[3] = s2.y;
      q[4] = s1.z;
      q[5] = s2.z;
      q[6] = s1.w;
      q[7] = s2.w;
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
};
