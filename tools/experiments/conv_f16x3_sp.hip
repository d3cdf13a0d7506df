// conv_f16x3 as ONE software-pipelined wave per SIMD: the same arithmetic, weight layout and epilogue as conv_f16x3.hip
// (read that file's header first), a different execution shape.
//
// Why (round 2 measurements; profiles/r02_pmc_canonical.json, tools/experiments/README.md): in conv_f16x3_kernel the two
// co-resident workgroups of a CU spend their matrix phases together (48 % of the wave cycles are stalls on the matrix
// pipe) and then stage / store together (the pipe idles: 43 % busy over the launch).  Splitting the roles over different
// waves did not help either — staging + epilogue is as much vector work as the matrix work, and two waves on a SIMD
// fight over its issue port.  What does overlap is ONE wave's own instruction stream: a v_mfma_f32_32x32x16_f16 keeps
// the matrix pipe busy for 32 cycles but the wave's issue for 8, so ~5 vector instructions fit behind each
// (tools/micro/coissue.hip).  So:
//
//   one persistent 4-wave workgroup per CU (one wave per SIMD, the whole 512-register file); each wave owns the
//   accumulators of 64 pixels x 64 output channels AND a quarter of the staging / epilogue work, and walks chunk
//   periods.  In period q the MFMAs of chunk q (out of LDS tile buffer q & 1) are interleaved, weight step by weight
//   step, with: the prologue + block maximum + fp16 split + LDS write of chunk q + 1 (into the other tile buffer), the
//   HBM prefetch of chunk q + 3 (two register sets, inline-asm loads with hand-counted vmcnt: see below), and the row
//   epilogue of the tile that has just ended (its accumulators parked in an LDS slab).
//
// Synchronisation: two workgroup barriers per period — M after the block maximum of chunk q + 1, E at the end — raw
// s_barrier behind an lgkmcnt(0) only.  The accumulator slab is private to its wave (written and drained by the same
// wave); its second half aliases the tile buffer the workgroup has just finished with (written after E, drained before
// the M of the next period, i.e. before that buffer is refilled).
//
// HBM prefetch: inline asm + hand-counted waits, because (1) vector-memory operations retire in order, (2) hipcc's own
// wait insertion falls back to vmcnt(0) in this control flow and would drain the chunk just requested, (3) vmcnt counts
// stores too — every store of a period is therefore issued right behind the period's one wait.
//
// Replaces (same C-ABI entry dmh_conv2d, same packed weights): the stride-1 3x3 launches of conv_f16x3.hip where it is
// faster (dmh_f16x3_sp_wanted).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#include "conv_args.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32;            // input channels per chunk
constexpr int PITCH = 160;        // LDS bytes per staged pixel (conv_f16x3.hip)
constexpr int STEP_U4 = 4 * 64;   // uint4 per (chunk, tap, cout half) of the packed weight
constexpr int EP = EpilogueRows::EP;

// LDS operations of this wave retired, then the workgroup barrier; nothing is said about vmcnt: loads stay in flight
#define SP_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int KH, int KW, int TH, int WM, int WN>
struct SpCfg {
  static_assert(WM * WN == 4 && TH * 16 == WM * 64, "four waves of 64 pixels");
  static constexpr int IN_H = TH + KH - 1, IN_W = 16 + KW - 1, IN_PIX = IN_H * IN_W;
  static constexpr int NLOAD = (IN_PIX * 8 + 255) / 256;
  static constexpr int PAD = KH / 2;
  // an A fragment of the 32x32x16 MFMA spans TWO tile rows (32 pixels); a 16-lane ds_read_b128 service group then pairs
  // 8 pixels of row y with 8 of row y + 1, whose 16 B slots (10 * px mod 16: all even) must differ in parity: odd row pitch
  static constexpr int ROWP = IN_W * PITCH + (((IN_W * PITCH / 16) % 2 == 0) ? 16 : 0);
  static constexpr int IN_BYTES = IN_H * ROWP;
  static constexpr int HALF_BYTES = 4 * 32 * EP * 4;          // 32 slab rows of each of the 4 waves
  static constexpr bool ALIAS = IN_BYTES >= HALF_BYTES;       // second slab half lives in the tile buffer just consumed
  static constexpr int SLAB0 = 2 * IN_BYTES;
  static constexpr int SLAB1 = SLAB0 + HALF_BYTES;            // (unused when ALIAS)
  static constexpr int RED = ALIAS ? SLAB1 : SLAB1 + HALF_BYTES;   // 4 waves x 64 channels x (sum, sum^2)
  static constexpr int SLOTS = RED + 4 * 64 * 2 * 4;
  static constexpr int LDS_BYTES = SLOTS + 16;
  static constexpr int NTAPS = KH * KW;
  static constexpr int TAPS_A = (NTAPS + 1) / 2;              // taps before barrier M
};

__device__ __forceinline__ unsigned absbits(float x) { return __float_as_uint(x) & 0x7fffffffu; }

// Rules for the hand-waited loads: (1) a register written by gload16 is read only behind vm_wait<N>() + pin(), with N =
// the number of vector-memory operations certainly issued after that load (more of them in flight only make the wait
// longer, never wrong); (2) every load of the kernel's steady state except the weight stream is of this kind.
__device__ __forceinline__ float4v gload16(const float* ptr) {
  float4v d;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(ptr) : "memory");
  return d;
}
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin(float4v& x) { asm volatile("" : "+v"(x)); }   // orders the uses of x behind vm_wait
// the same with the destination in the ACCUMULATOR half of the register file: a load in flight for a whole period (the
// chunk prefetch, the tile constants) must not occupy one of the 256 architectural VGPRs the vector instructions need;
// the compiler copies to a VGPR (v_accvgpr_read) where a vector instruction reads it — behind pin_a, i.e. behind the wait
__device__ __forceinline__ float4v gload16a(const float* ptr) {
  float4v d;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(d) : "v"(ptr) : "memory");
  return d;
}
__device__ __forceinline__ void pin_a(float4v& x) { asm volatile("" : "+a"(x)); }

template <int V>
using IC = std::integral_constant<int, V>;

// compile-time loop: f(IC<0>{}), ..., f(IC<N-1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_seq(f, std::make_integer_sequence<int, N>{});
}

}  // namespace

// PRO: the launch has a GroupNorm + SiLU prologue on source 0 (p.in_coef)
template <int KH, int KW, int TH, int WM, int WN, bool PRO>
__global__ __launch_bounds__(256, 1) void conv_f16x3_sp_kernel(ConvArgs p, int ntiles, int ups) {
  using Cfg = SpCfg<KH, KW, TH, WM, WN>;
  constexpr int IN_W = Cfg::IN_W, IN_PIX = Cfg::IN_PIX, NLOAD = Cfg::NLOAD, NTAPS = Cfg::NTAPS, ROWP = Cfg::ROWP;
  constexpr int NSTEP = NTAPS * 2;                  // weight steps per chunk: (tap, 32-column block of the wave's 64 channels)
  constexpr int NS_A = Cfg::TAPS_A * 2, NS_B = NSTEP - NS_A;   // steps before / after barrier M
  constexpr int NB = (NSTEP % 3 == 0) ? 3 : 2;      // rotating B buffers: loads run NB - 1 steps ahead
  // EVERY load of the steady state is a hand-waited one (gload16), the weight stream included — one load the compiler
  // waits for by itself would count only its own kind and drain the HBM prefetch sitting between them in the queue.
  // In flight when a period's first wait runs, oldest first: [chunk q + 1] [weight steps 0, 1] [tile constants] [chunk q + 2]
  constexpr int KEEP = NLOAD + 2;                   // ... of which the NLOAD + 2 loads of chunk q + 2 stay in flight
  constexpr int KEEP_B = 4 * (NB - 1);              // at weight step s >= NB - 1: the steps s + 1 .. s + NB - 1 stay in flight

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* slot = reinterpret_cast<unsigned*>(smem + Cfg::SLOTS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- this workgroup's tiles: XCD k (= blockIdx.x & 7 under round-robin placement; speed only) walks the k-th
  // contiguous run of (cout tile, sample, tile row, tile column), its workgroups interleaved inside the run
  const int G8 = gridDim.x >> 3, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int rq = ntiles >> 3, rr = ntiles & 7;
  const int run0 = xcd * rq + min(xcd, rr), runlen = rq + (xcd < rr ? 1 : 0);
  const int T = local < runlen ? (runlen - local + G8 - 1) / G8 : 0;   // tiles of this workgroup
  if (T == 0) return;
  const int nch = p.nch0 + p.nch1;                  // >= 2 (the slab needs a period between two tile ends)
  const int Q = T * nch;                            // chunk periods of this workgroup
  auto tile_coords = [&](int j, int& ct, int& b, int& ty, int& tx) __attribute__((always_inline)) {
    int L = run0 + local + j * G8;
    tx = L % p.tilesX;
    L /= p.tilesX;
    ty = L % p.tilesY;
    L /= p.tilesY;
    b = L % p.B;
    ct = L / p.B;
  };

  if (tid < 2) slot[tid] = 0u;
  __syncthreads();

  // =====================================================================================================================
  //                                          staging state (a quarter of the workgroup's halo tile per wave)
  // =====================================================================================================================
  const int c4 = tid & 7;
  float4v v[2][NLOAD];                // two halo-tile chunks in flight (gload16)
  float4v ca[2], cb[2];
  unsigned inside_set[2] = {0u, 0u};  // validity mask of each set's pixels
  int poff[NLOAD], wroff[NLOAD];
  unsigned inside_lt = 0u;            // ... of the tile whose loads are being issued
  int lt_j = -1, lt_b = 0;
  int e_run_p = 16;
  unsigned mx_p = 0u;                 // running |x| maximum of the chunk being staged (this thread)
  float sc_p = 1.f;                   // its block scale
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int pix = (tid + i * 256) >> 3;
    const int pixc = pix < IN_PIX ? pix : IN_PIX - 1;
    wroff[i] = (pixc / IN_W) * ROWP + (pixc % IN_W) * PITCH + c4 * 8;
    poff[i] = 0;
  }
  const int Hlim = ups ? p.Hin * 2 : p.Hin, Wlim = ups ? p.Win * 2 : p.Win;
  auto set_load_tile = [&](int j) __attribute__((always_inline)) {
    int ct, b, ty, tx;
    tile_coords(j, ct, b, ty, tx);
    const int iy0 = ty * TH - Cfg::PAD, ix0 = tx * 16 - Cfg::PAD;
    inside_lt = 0u;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int pix = (tid + i * 256) >> 3;
      const int pixc = pix < IN_PIX ? pix : IN_PIX - 1;
      const int yy = iy0 + pixc / IN_W, xx = ix0 + pixc % IN_W;
      const bool ok = pix < IN_PIX && yy >= 0 && yy < Hlim && xx >= 0 && xx < Wlim;
      const int yc = min(max(yy, 0), Hlim - 1), xc = min(max(xx, 0), Wlim - 1);
      const int sy = ups ? (yc >> 1) : yc, sx = ups ? (xc >> 1) : xc;
      poff[i] = (b * p.Hin + sy) * p.Win + sx;
      inside_lt |= (ok ? 1u : 0u) << i;
    }
    lt_j = j;
    lt_b = b;
  };
  // loads of chunk period qq (clamped to the last one: a harmless re-load) into register set SET: NLOAD + 2 operations
  auto issue_loads = [&](auto SETC, int qq) __attribute__((always_inline)) {
    constexpr int SET = decltype(SETC)::value;
    qq = qq < Q ? qq : Q - 1;
    const int j = qq / nch, ch = qq - j * nch;
    if (j != lt_j) set_load_tile(j);
    const bool s1 = ch >= p.nch0;
    const float* src = s1 ? p.src1 : p.src0;
    const int Csrc = s1 ? p.C1 : p.C0;
    const int c = (s1 ? ch - p.nch0 : ch) * KC + c4 * 4;
    const int cc = c < Csrc ? c : 0;
    const bool pro = PRO && !s1;
    const float* cf = pro ? p.in_coef : p.src0;             // (a valid address either way: the load is unconditional)
    const size_t o0 = pro ? (size_t)(lt_b * 2 + 0) * p.C0 + cc : 0, o1 = pro ? (size_t)(lt_b * 2 + 1) * p.C0 + cc : 0;
    ca[SET] = gload16a(cf + o0);
    cb[SET] = gload16a(cf + o1);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) v[SET][i] = gload16a(src + (size_t)poff[i] * Csrc + cc);
    inside_set[SET] = c < Csrc ? inside_lt : 0u;
  };
  // prologue in place + |x| maximum of float4 I of the set (chunk qn); straight-line (selects only)
  auto stage_max_one = [&](auto SETC, auto IC_, bool pro) __attribute__((always_inline)) {
    constexpr int SET = decltype(SETC)::value, I = decltype(IC_)::value;
    const bool ok = (inside_set[SET] >> I) & 1u;
    float4v x = v[SET][I];
    if (PRO) {
      // (pro is false for the chunks of a second source: PRO launches have none — dmh_conv2d refuses the combination)
      const float4v a4 = ca[SET], b4 = cb[SET];
      x.x = silu_fast(fmaf(a4.x, x.x, b4.x));
      x.y = silu_fast(fmaf(a4.y, x.y, b4.y));
      x.z = silu_fast(fmaf(a4.z, x.z, b4.z));
      x.w = silu_fast(fmaf(a4.w, x.w, b4.w));
    }
    (void)pro;
    x.x = ok ? x.x : 0.f;                            // padding is exactly zero: it pads the ACTIVATED tensor
    x.y = ok ? x.y : 0.f;
    x.z = ok ? x.z : 0.f;
    x.w = ok ? x.w : 0.f;
    v[SET][I] = x;
    mx_p = max(max(mx_p, absbits(x.x)), max(max(absbits(x.y), absbits(x.z)), absbits(x.w)));
  };
  auto stage_max_finish = [&](int qn) __attribute__((always_inline)) {
    unsigned mx = mx_p;
#pragma unroll
    for (int off = 32; off; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
    if (lane == 0) atomicMax(&slot[qn & 1], mx);
    mx_p = 0u;
  };
  auto stage_scale = [&](int qn) __attribute__((always_inline)) {
    const int ch = qn % nch;
    const unsigned bmx = slot[qn & 1];
    if (ch == 0) e_run_p = 16;
    const int e_ch = min(max((int)(__builtin_amdgcn_readfirstlane(bmx) >> 23), 16), 254);
    e_run_p = max(e_run_p, e_ch);
    sc_p = __uint_as_float((unsigned)(268 - e_run_p) << 23);
  };
  auto stage_write_one = [&](auto SETC, auto IC_, int qn) __attribute__((always_inline)) {
    constexpr int SET = decltype(SETC)::value, I = decltype(IC_)::value;
    if ((I + 1) * 256 <= IN_PIX * 8 || ((tid + I * 256) >> 3) < IN_PIX) {
      unsigned char* in_tile = smem + (qn & 1) * Cfg::IN_BYTES;
      const float4v xs = v[SET][I] * sc_p;
      const half4 h1 = __builtin_convertvector(xs, half4);
      const float4v rs = (xs - __builtin_convertvector(h1, float4v)) * 2048.f;
      const half4 h2 = __builtin_convertvector(rs, half4);
      unsigned char* dst = in_tile + wroff[I];
      *reinterpret_cast<half4*>(dst) = h1;
      *reinterpret_cast<half4*>(dst + 64) = h2;
    }
  };

  // =====================================================================================================================
  //                                          epilogue state (this wave's own 64 x 64 slab)
  // =====================================================================================================================
  int ep_ct = 0, ep_b = 0, ep_ty = 0, ep_tx = 0, ep_half1 = 0;
  bool ep_pending = false, st_pending = false;
  int st_ct = 0, st_b = 0, st_ty = 0, st_tx = 0;
  float4v ep_bias = float4v{0.f, 0.f, 0.f, 0.f}, ep_osc = float4v{1.f, 1.f, 1.f, 1.f};
  float4v ep_ra = ep_bias, ep_rb = ep_bias;           // raw loads (gload16), selected by ep_fix_consts
  bool ep_cok = false;
  float4 ep_s1 = make_float4(0.f, 0.f, 0.f, 0.f), ep_s2 = ep_s1;
  const int wm = wave / WN, wn = wave % WN;
  // constants of the tile that ends with this period: issued BEFORE the next prefetch (4 operations)
  auto ep_load_consts = [&](int j) __attribute__((always_inline)) {
    tile_coords(j, ep_ct, ep_b, ep_ty, ep_tx);
    const int chn = (ep_ct * WN + wn) * 64 + (lane & 15) * 4;
    const bool cok = chn < p.Cout;
    const int chc = cok ? chn : 0;
    ep_cok = cok;
    ep_osc = gload16a((p.oscale ? p.oscale : p.wpack) + chc);
    ep_bias = gload16a((p.bias ? p.bias : p.wpack) + chc);
    ep_ra = gload16a(p.res_coef ? p.res_coef + (size_t)(ep_b * 2 + 0) * p.Cout + chc : p.wpack);
    ep_rb = gload16a(p.res_coef ? p.res_coef + (size_t)(ep_b * 2 + 1) * p.Cout + chc : p.wpack);
  };
  auto ep_fix_consts = [&]() __attribute__((always_inline)) {
    pin_a(ep_osc);
    pin_a(ep_bias);
    pin_a(ep_ra);
    pin_a(ep_rb);
    const float4v zero4 = float4v{0.f, 0.f, 0.f, 0.f}, one4 = float4v{1.f, 1.f, 1.f, 1.f};
    ep_osc = (ep_cok && p.oscale) ? ep_osc : one4;
    ep_bias = (ep_cok && p.bias) ? ep_bias : zero4;
    ep_ra = (ep_cok && p.res_coef) ? ep_ra : zero4;
    ep_rb = (ep_cok && p.res_coef) ? ep_rb : zero4;
  };
  // slab row group R (0..15: 4 rows each; 0-7 first half, 8-15 second half) -> NHWC rows
  auto ep_row = [&](int R) __attribute__((always_inline)) {
    const int hb = R >> 3, i = R & 7;
    const unsigned char* half_base = smem + (hb == 0 ? Cfg::SLAB0 : ep_half1);
    const float* wl = reinterpret_cast<const float*>(half_base) + wave * (32 * EP);
    const int c4e = lane & 15, rsub = lane >> 4;
    const int chn = (ep_ct * WN + wn) * 64 + c4e * 4;
    const bool cok = chn < p.Cout;
    const int rrow = i * 4 + rsub;
    const int row = wm * 64 + hb * 32 + rrow;
    const int oy = ep_ty * TH + row / 16, ox = ep_tx * 16 + row % 16;
    if (cok && oy < p.Hout && ox < p.Wout) {
      float4 val = ld4(wl + rrow * EP + c4e * 4);
      const size_t o = ((size_t)(ep_b * p.Hout + oy) * p.Wout + ox) * p.Cout + chn;
      val.x = fmaf(val.x, ep_osc.x, ep_bias.x);
      val.y = fmaf(val.y, ep_osc.y, ep_bias.y);
      val.z = fmaf(val.z, ep_osc.z, ep_bias.z);
      val.w = fmaf(val.w, ep_osc.w, ep_bias.w);
      if (p.res) {
        const float4 rv = ld4(p.res + o);
        if (p.res_coef) {
          val.x += silu_fast(fmaf(ep_ra.x, rv.x, ep_rb.x));
          val.y += silu_fast(fmaf(ep_ra.y, rv.y, ep_rb.y));
          val.z += silu_fast(fmaf(ep_ra.z, rv.z, ep_rb.z));
          val.w += silu_fast(fmaf(ep_ra.w, rv.w, ep_rb.w));
        } else {
          val.x += rv.x;
          val.y += rv.y;
          val.z += rv.z;
          val.w += rv.w;
        }
      }
      st4(p.out + o, val);
      ep_s1.x += val.x;
      ep_s1.y += val.y;
      ep_s1.z += val.z;
      ep_s1.w += val.w;
      ep_s2.x = fmaf(val.x, val.x, ep_s2.x);
      ep_s2.y = fmaf(val.y, val.y, ep_s2.y);
      ep_s2.z = fmaf(val.z, val.z, ep_s2.z);
      ep_s2.w = fmaf(val.w, val.w, ep_s2.w);
    }
  };
  // this wave's GroupNorm partials -> LDS (combined across waves behind the next period's wait, two barriers later)
  auto ep_stats_partial = [&]() __attribute__((always_inline)) {
    if (p.stats) {
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        ep_s1.x += __shfl_xor(ep_s1.x, off);
        ep_s1.y += __shfl_xor(ep_s1.y, off);
        ep_s1.z += __shfl_xor(ep_s1.z, off);
        ep_s1.w += __shfl_xor(ep_s1.w, off);
        ep_s2.x += __shfl_xor(ep_s2.x, off);
        ep_s2.y += __shfl_xor(ep_s2.y, off);
        ep_s2.z += __shfl_xor(ep_s2.z, off);
        ep_s2.w += __shfl_xor(ep_s2.w, off);
      }
      float* red = reinterpret_cast<float*>(smem + Cfg::RED);
      if (lane < 16) {
        float* q = red + (wave * 64 + (lane & 15) * 4) * 2;
        q[0] = ep_s1.x;
        q[1] = ep_s2.x;
        q[2] = ep_s1.y;
        q[3] = ep_s2.y;
        q[4] = ep_s1.z;
        q[5] = ep_s2.z;
        q[6] = ep_s1.w;
        q[7] = ep_s2.w;
      }
    }
    ep_s1 = make_float4(0.f, 0.f, 0.f, 0.f);
    ep_s2 = ep_s1;
    st_ct = ep_ct;
    st_b = ep_b;
    st_ty = ep_ty;
    st_tx = ep_tx;
    st_pending = true;
  };
  // stat tiles are 8 rows x 16 columns = the 128 pixels of two waves along M (conv_args.h write_stats_grid)
  auto ep_stats_combine = [&]() __attribute__((always_inline)) {
    st_pending = false;
    if (!p.stats || tid >= 128) return;
    const float* red = reinterpret_cast<const float*>(smem + Cfg::RED);
    const int q = tid >> 6, c = tid & 63;
    const int g = WN == 1 ? q : 0, wnq = WN == 1 ? 0 : q;
    const int chan = (st_ct * WN + wnq) * 64 + c;
    const int srow = st_ty * (TH / 8) + g;
    if (chan < p.Cout && srow * 8 < p.Hout) {
      const int w0 = (g * 2 + 0) * WN + wnq, w1 = (g * 2 + 1) * WN + wnq;
      const float a0 = red[(w0 * 64 + c) * 2 + 0] + red[(w1 * 64 + c) * 2 + 0];
      const float a1 = red[(w0 * 64 + c) * 2 + 1] + red[(w1 * 64 + c) * 2 + 1];
      const int stiles = ((p.Hout + 7) / 8) * p.tilesX;
      float* st = p.stats + ((size_t)(st_b * stiles + srow * p.tilesX + st_tx) * p.Cout + chan) * 2;
      st[0] = a0;
      st[1] = a1;
    }
  };

  // =====================================================================================================================
  //                                          matrix state: v_mfma_f32_32x32x16_f16
  // =====================================================================================================================
  // A wave's 64 pixels are two 32-row blocks (two tile rows each), its 64 output channels two 32-column blocks; a chunk's
  // K = 32 is two K steps of 16.  Fragment maps: lane l, r = l & 31, h = l >> 5: A[row r][k = 8h + j], B[k = 8h + j][col r];
  // C: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).  The packed weight keeps its 16x16x32 fragment order
  // [cout half][16-col block][plane][k group * 16 + col][8]: a 32x32x16 B fragment is read from it with per-lane addresses.
  const int r31 = lane & 31, hh = lane >> 5;
  int arow[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) arow[mb] = (wm * 4 + mb * 2 + (r31 >> 4)) * ROWP + (r31 & 15) * PITCH + hh * 16;
  int boff[2];                                        // uint4 offset of this lane's B slice for K step ks (plane g1; g2: + 64)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) boff[ks] = (((r31 >> 4) & 1) * 2) * 64 + (2 * ks + hh) * 16 + (r31 & 15);
  floatx16 acc[2][2];
  float4v bq[NB][4];                                  // [K step][plane] (gload16)
  half8 a[2][2][2][2];                                // [buffer][32-row block][K step][plane]
  int e_run = 16;
  const int nsteps = nch * NSTEP;                     // weight steps per tile
  auto wbase_of = [&](int j) __attribute__((always_inline)) {
    int ct, b, ty, tx;
    tile_coords(j < T ? j : T - 1, ct, b, ty, tx);
    return reinterpret_cast<const uint4*>(p.wpack) + (size_t)(ct * WN + wn) * nsteps * STEP_U4;
  };
  const uint4* wcur = wbase_of(0);
  const uint4* wnxt = wbase_of(1);
  auto load_b = [&](int buf, int step) __attribute__((always_inline)) {   // 4 loads; step counts from the tile start
    const uint4* src = step < nsteps ? wcur + (size_t)step * STEP_U4 : wnxt + (size_t)(step - nsteps) * STEP_U4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bq[buf][ks * 2 + pl] = gload16(reinterpret_cast<const float*>(src + boff[ks] + pl * 64));
  };
  auto read_a = [&](int ab, const unsigned char* in_tile, int tap) __attribute__((always_inline)) {
    const unsigned char* at = in_tile + (tap / KW) * ROWP + (tap % KW) * PITCH;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          a[ab][mb][ks][pl] = *reinterpret_cast<const half8*>(at + arow[mb] + ks * 32 + pl * 64);
  };
  int step = 0;
  // one weight step ST of the chunk staged in in_tile: 24 MFMAs
  auto mma_step = [&](auto STC, const unsigned char* in_tile) __attribute__((always_inline)) {
    constexpr int st = decltype(STC)::value;          // st = 2 * tap + (32-column block)
    load_b((st + NB - 1) % NB, step + NB - 1);
    // weight steps 0 .. NB - 2 of a period are older than the chunk prefetch issued at the end of the previous period and
    // came in with the period's first wait; from then on: leave the NB - 1 steps ahead in flight
    if (st >= NB - 1) vm_wait<KEEP_B>();
#pragma unroll
    for (int i = 0; i < 4; ++i) pin(bq[st % NB][i]);
    constexpr int tap = st >> 1, cbk = st & 1;
    if (cbk == 0 && tap + 1 < NTAPS) read_a((tap + 1) & 1, in_tile, tap + 1);
    half8 g1s[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) g1s[ks] = __builtin_bit_cast(half8, bq[st % NB][ks * 2]) * (_Float16)(1.0f / 2048.0f);
    __builtin_amdgcn_sched_barrier(0);                // (keeps the wait + the loads of the steps ahead in front of the MFMAs)
#define DMH_TERM(pl, bexpr)                                                                       \
  _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      acc[mb][cbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap & 1][mb][ks][pl], bexpr, acc[mb][cbk], 0, 0, 0);
    DMH_TERM(1, g1s[ks])                                                 // h2 * g1s   (smallest terms first)
    DMH_TERM(0, __builtin_bit_cast(half8, bq[st % NB][ks * 2 + 1]))      // h1 * g2
    DMH_TERM(0, __builtin_bit_cast(half8, bq[st % NB][ks * 2]))          // h1 * g1
#undef DMH_TERM
    ++step;
  };
  // scheduling pattern of one weight-step region (its 24 MFMAs + the staging piece that follows them in the source): one
  // MFMA, then up to three vector instructions in its shadow (the MFMA holds the issue port for 8 of its 32 cycles)
  auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
  };
  auto slab_write = [&](int hb, unsigned char* half_base, float inv_s) __attribute__((always_inline)) {
    float* wl = reinterpret_cast<float*>(half_base) + wave * (32 * EP);
#pragma unroll
    for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        wl[((r & 3) + 8 * (r >> 2) + 4 * hh) * EP + cbk * 32 + r31] = acc[hb][cbk][r] * inv_s;
  };

  // =====================================================================================================================
  // period -1: chunk 0 of the first tile is staged (nothing to overlap it with), the weight stream starts
  // =====================================================================================================================
  issue_loads(IC<0>{}, 0);
  issue_loads(IC<1>{}, 1);
#pragma unroll
  for (int i = 0; i < NB - 1; ++i) load_b(i, i);
  vm_wait<KEEP>();
  pin_a(ca[0]);
  pin_a(cb[0]);
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) pin_a(v[0][i]);
  static_for<NLOAD>([&](auto I) __attribute__((always_inline)) { stage_max_one(IC<0>{}, I, PRO); });
  stage_max_finish(0);
  SP_BAR();
  stage_scale(0);
  static_for<NLOAD>([&](auto I) __attribute__((always_inline)) { stage_write_one(IC<0>{}, I, 0); });
  issue_loads(IC<0>{}, 2);
  SP_BAR();

  // =====================================================================================================================
  // one chunk period q.  SETN = the register set holding chunk q + 1; EPT = the previous tile's rows are pending
  // =====================================================================================================================
  auto period = [&](auto SETN, int q) __attribute__((always_inline)) {
    constexpr int SET = decltype(SETN)::value;
    const int j = q / nch, ch = q - j * nch;
    const bool last_ch = ch == nch - 1;
    unsigned char* buf = smem + (q & 1) * Cfg::IN_BYTES;
    if (ch == 0) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][cbk][r] = 0.f;
      e_run = 16;
    }
    // ---- block scale of chunk q: running maximum over the tile's chunks, so the scale only ever shrinks
    {
      const unsigned bmx = slot[q & 1];
      const int e_old = e_run;
      const int e_ch = min(max((int)(__builtin_amdgcn_readfirstlane(bmx) >> 23), 16), 254);
      e_run = max(e_run, e_ch);
      if (e_run != e_old && ch > 0) {
        const int fe = 127 + e_old - e_run;
        const float f = fe > 0 ? __uint_as_float((unsigned)fe << 23) : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int cbk = 0; cbk < 2; ++cbk) acc[mb][cbk] *= f;
      }
    }
    read_a(0, buf, 0);
    // ---- the period's one wait.  In flight, oldest first: [chunk q + 1] [tile constants] [chunk q + 2] [weight steps]:
    // everything of a period that STORES is issued behind it and is a whole period old at the next wait
    vm_wait<KEEP>();
    pin_a(ca[SET]);
    pin_a(cb[SET]);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) pin_a(v[SET][i]);
    if (st_pending) ep_stats_combine();
    if (ep_pending) {
      // the previous tile's rows: a plain loop in front of the period's matrix work (interleaving its 16 row groups with
      // the weight steps cost > 100 spilled registers under hipcc; it is ~10 % of the period's vector work)
      ep_fix_consts();
#pragma clang loop unroll(disable)
      for (int R = 0; R < 16; ++R) ep_row(R);
      ep_stats_partial();
      ep_pending = false;
    }
    // ---- first half: steps [0, NS_A) || rows of the previous tile, prologue + maximum of chunk q + 1
    static_for<NS_A>([&](auto KC_) __attribute__((always_inline)) {
      constexpr int K_ = decltype(KC_)::value;
      mma_step(IC<K_>{}, buf);
      static_for<(K_ + 1) * NLOAD / NS_A - K_ * NLOAD / NS_A>([&](auto I) __attribute__((always_inline)) {
        stage_max_one(SETN, IC<K_ * NLOAD / NS_A + decltype(I)::value>{}, PRO);
      });
      interleave();
    });
    stage_max_finish(q + 1);
    SP_BAR();  // M: block maximum of chunk q + 1 complete
    stage_scale(q + 1);
    if (tid == 0) slot[q & 1] = 0u;                  // (everybody has read it)
    // ---- second half: steps [NS_A, NSTEP) || split + LDS write of chunk q + 1
    if constexpr (NS_B > 0) {
      static_for<NS_B>([&](auto KC_) __attribute__((always_inline)) {
        constexpr int K_ = decltype(KC_)::value;
        mma_step(IC<NS_A + K_>{}, buf);
        static_for<(K_ + 1) * NLOAD / NS_B - K_ * NLOAD / NS_B>([&](auto I) __attribute__((always_inline)) {
          stage_write_one(SETN, IC<K_ * NLOAD / NS_B + decltype(I)::value>{}, q + 1);
        });
        interleave();
      });
    } else {
      static_for<NLOAD>([&](auto I) __attribute__((always_inline)) { stage_write_one(SETN, I, q + 1); });
    }
    if (last_ch) ep_load_consts(j);
    issue_loads(SETN, q + 3);
    const float inv_s = __uint_as_float((unsigned)(e_run - 14) << 23);  // 1 / block scale
    if (last_ch) slab_write(0, smem + Cfg::SLAB0, inv_s);
    SP_BAR();  // E: tile buffer (q + 1) & 1 is staged; everybody is done reading buffer q & 1
    if (last_ch) {
      ep_half1 = Cfg::ALIAS ? (q & 1) * Cfg::IN_BYTES : Cfg::SLAB1;
      slab_write(1, smem + ep_half1, inv_s);
      ep_pending = true;
      step = 0;
      wcur = wnxt;
      wnxt = wbase_of(j + 2);
    }
  };
  for (int q = 0; q < Q; q += 2) {
    period(IC<1>{}, q);
    if (q + 1 < Q) period(IC<0>{}, q + 1);
  }
  // ---- tail: the last tile's rows
  vm_wait<0>();
  if (st_pending) ep_stats_combine();
  if (ep_pending) {
    ep_fix_consts();
#pragma clang loop unroll(disable)
    for (int R = 0; R < 16; ++R) ep_row(R);
    ep_stats_partial();
  }
  SP_BAR();
  if (st_pending) ep_stats_combine();
}

// ------------------------------------------------------------------------------ host side
static int sp_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int KH, int KW, int TH, int WM, int WN, bool PRO>
static int launch_sp(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = SpCfg<KH, KW, TH, WM, WN>;
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, 16);
  a.oscale = d->wpack + dmh_f16x3_pack_floats(d->Cout, a.C0, a.C1, KH, KW) - (int64_t)cdiv(d->Cout, 64) * 64;
  auto kern = conv_f16x3_sp_kernel<KH, KW, TH, WM, WN, PRO>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Cfg::LDS_BYTES);
    DMH_REQUIRE(e == hipSuccess, "dmh_conv2d: cannot raise the LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int ntiles = a.tilesX * a.tilesY * a.B * cdiv(a.Cout, 64 * WN);
  int grid = sp_num_cus() & ~7;                      // one persistent workgroup per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, st, a, ntiles, d->upsample2 == 1 ? 1 : 0);
  DMH_CHECK_LAUNCH("dmh_conv2d(f16x3 software-pipelined)");
  return DMH_OK;
}

// Which launches take this kernel (DMH_CONV_SP: 0 = never, 1 = the measured default, 2 = wherever it can run)
bool dmh_f16x3_sp_wanted(const DmhConv* d, int Hout, int Wout) {
  static const int mode = [] {
    const char* e = getenv("DMH_CONV_SP");
    return e ? atoi(e) : 0;
  }();
  if (mode == 0) return false;
  const int C1 = d->src1 ? d->C1 : 0;
  const int nch = cdiv(d->C0, KC) + cdiv(C1, KC);
  if (!(d->stride == 1 && d->KH == 3 && d->KW == 3 && d->upsample2 == 0 && nch >= 2)) return false;
  (void)Hout;
  (void)Wout;
  return true;
}

int dmh_f16x3_sp_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  const bool wide = d->Cout % 128 == 0;
  if (d->in_coef)
    return wide ? launch_sp<3, 3, 8, 2, 2, true>(d, Hout, Wout, st) : launch_sp<3, 3, 16, 4, 1, true>(d, Hout, Wout, st);
  return wide ? launch_sp<3, 3, 8, 2, 2, false>(d, Hout, Wout, st) : launch_sp<3, 3, 16, 4, 1, false>(d, Hout, Wout, st);
}
