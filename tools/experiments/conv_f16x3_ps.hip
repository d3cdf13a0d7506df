// conv_f16x3 with the waves of a workgroup SPECIALISED: four consumer waves that only issue matrix instructions and four
// producer waves that do everything else — the same arithmetic, weight layout and epilogue as conv_f16x3.hip (read that
// file's header first), a different execution shape.
//
// Why (measured, round 2: tools/f16_ablate.py on 64->64 @128^2, B = 50): in conv_f16x3_kernel every wave walks
// load -> prologue -> split -> LDS -> matrix -> epilogue, and the phases of the two co-resident workgroups ADD UP instead of
// overlapping: 108 of 242 us remain with the matrix phase removed, the matrix pipe is busy < 50 % of the cycles, and the
// HBM traffic happens in bursts (a tile's 83 KB in, its 65 KB out) during which nothing multiplies.  A CU has to move
// ~150 KB per tile at the ~24 GB/s that is its share of HBM — about as long as the tile's 13.8 k matrix cycles — so both
// have to run ALL the time:
//
//   one persistent workgroup per CU = 8 waves: waves 0-3 ("consumers", one per SIMD) own the accumulators of a
//   TH x 16 pixel x (64 * WN) channel tile and do nothing but ds_read A fragments / stream B fragments / MFMA;
//   waves 4-7 ("producers", the SIMD partners of 0-3) keep two halo-tile chunks in flight from HBM (two register
//   sets, issued 1.5 chunk periods ahead), apply the fused SiLU(a*x+b) prologue, find the block maximum, split into
//   the two fp16 planes and write them into the OTHER of two LDS tile buffers while the consumers multiply out of the
//   first; they also take over the finished tile's accumulators through an LDS slab and do the whole row epilogue
//   (+bias, +residual, GroupNorm partials, float4 NHWC stores) during the next tile's matrix phase.
//
// Synchronisation: two workgroup barriers per chunk period (M after the producers' block maximum, E at the end) plus one
// (S) per tile for the slab hand-over — raw s_barrier behind an lgkmcnt(0) only, so that neither the producers' HBM
// prefetch nor the consumers' weight prefetch is drained at a barrier.  Every thread executes the same barrier sequence.
//
// LDS (4x1 consumers, 16 x 16 tile): 2 x 51,840 B tile buffers + 34,816 B for the first half of the accumulator slab;
// the second half (rows 32-63 of every consumer) is written INTO the tile buffer the consumers have just finished with
// (after barrier E) and is drained by the producers before they refill that buffer (before barrier M of the next period).
// 2x2 consumers (8 x 16 tile, 128 channels): 2 x 28,800 B + the whole slab.
//
// Replaces (same C-ABI entry dmh_conv2d, same packed weights): the stride-1 3x3 / 1x1 launches of conv_f16x3.hip where
// it is faster (dmh_f16x3_ps_wanted).
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

// LDS writes of this wave retired, then the workgroup barrier; nothing is said about vmcnt: loads stay in flight
#define PS_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int KH, int KW, int TH, int WM, int WN, int MF>
struct PsCfg {
  static_assert(WM * WN == 4 && TH * 16 == WM * 64, "four consumer waves of 64 pixels");
  static_assert(MF == 16 || MF == 32, "v_mfma_f32_16x16x32_f16 or v_mfma_f32_32x32x16_f16");
  static constexpr int IN_H = TH + KH - 1, IN_W = 16 + KW - 1, IN_PIX = IN_H * IN_W;
  static constexpr int NLOAD = (IN_PIX * 8 + 255) / 256;
  static constexpr int PAD = KH / 2;
  // MF == 32: an A fragment spans TWO tile rows (32 pixels); a 16-lane ds_read_b128 service group then pairs 8 pixels of
  // row y with 8 of row y + 1, whose 16 B slots (10 * px mod 16: all even) must differ in parity: row pitch / 16 odd
  static constexpr int ROWP = IN_W * PITCH + (MF == 32 && ((IN_W * PITCH / 16) % 2 == 0) ? 16 : 0);
  static constexpr int IN_BYTES = IN_H * ROWP;
  static constexpr int HALF_BYTES = 4 * 32 * EP * 4;          // 32 slab rows of each of the 4 consumers
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

// The producers' HBM prefetch is issued through inline asm and waited for BY HAND.  Vector-memory operations retire in
// order and hipcc's own wait insertion falls back to vmcnt(0) in this kernel's control flow (seen in the ISA: every
// period then drained the chunk that had just been requested); loads the compiler does not see are loads it cannot
// wait for.  Rules kept by the code below: (1) a register written by gload16 is read only behind vm_wait<N>() + pin(),
// with N = the number of vector-memory operations certainly issued after that load (more of them in flight only make
// the wait longer, never wrong); (2) every producer-side load is of this kind, so no compiler-inserted wait drains them.
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

}  // namespace

#ifdef DMH_STAMPS
// diagnostic build only (make stamps; tools/ps_stamps.py): per-wave cycle totals by phase -> a buffer of the tool's
static unsigned long long* g_ps_dbg = nullptr;
extern "C" void dmh_ps_set_debug_buffer(void* p) { g_ps_dbg = reinterpret_cast<unsigned long long*>(p); }
#define PS_STAMP(i)                                                              \
  __builtin_amdgcn_sched_barrier(0);                                             \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory"); \
  __builtin_amdgcn_sched_barrier(0);                                             \
  tk[i] += t_now - t_prev;                                                       \
  t_prev = t_now;
#else
#define PS_STAMP(i)
#endif

template <int KH, int KW, int TH, int WM, int WN, int MF>
__global__ __launch_bounds__(512, 2) void conv_f16x3_ps_kernel(ConvArgs p, int ntiles, int ups
#ifdef DMH_STAMPS
                                                                 , unsigned long long* dbg
#endif
) {
  using Cfg = PsCfg<KH, KW, TH, WM, WN, MF>;
  constexpr int IN_W = Cfg::IN_W, IN_PIX = Cfg::IN_PIX, NLOAD = Cfg::NLOAD, NTAPS = Cfg::NTAPS, ROWP = Cfg::ROWP;
  constexpr int NSTEP = NTAPS * 2;                  // weight steps per chunk: (tap, half of the wave's 64 output channels)
  constexpr int NB = (NSTEP % 3 == 0) ? 3 : 2;      // rotating B buffers: loads run NB - 1 steps ahead
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

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
  const int nch = p.nch0 + p.nch1;                  // >= 2 (the slab hand-over needs a period between two tile ends)
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
#ifdef DMH_STAMPS
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev, t_now;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  const unsigned long long t_begin = t_prev;
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif

  if (wave >= 4) {
    // ===================================================================================================================
    //                                                     PRODUCERS
    // ===================================================================================================================
    const int ptid = tid & 255;
    const int c4 = ptid & 7;
    float4v v[2][NLOAD];                // two halo-tile chunks in flight (gload16: see the rules above)
    float4v ca[2], cb[2];
    unsigned inside_set[2] = {0u, 0u};  // validity mask of each set's pixels
    int poff[NLOAD], wroff[NLOAD];
    unsigned inside_lt = 0u;            // ... of the tile whose loads are being issued
    int lt_j = -1, lt_b = 0;
    int e_run_p = 16;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int pix = (ptid + i * 256) >> 3;
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
        const int pix = (ptid + i * 256) >> 3;
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
    // loads of chunk period qq (clamped to the last one: a harmless re-load) into register set SET.  Unconditional,
    // from clamped addresses, so that hipcc keeps counted vmcnt waits (conv.hip)
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
      // the two coefficient rows go first: vector-memory operations retire in order, so whoever waits for a pixel of
      // this set has them too (no second, conservative wait inside the prologue)
      const bool pro = p.in_coef != nullptr && !s1;
      const float* cf = pro ? p.in_coef : p.src0;           // (a valid address either way: the load is unconditional)
      const size_t o0 = pro ? (size_t)(lt_b * 2 + 0) * p.C0 + cc : 0, o1 = pro ? (size_t)(lt_b * 2 + 1) * p.C0 + cc : 0;
      ca[SET] = gload16(cf + o0);
      cb[SET] = gload16(cf + o1);
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) v[SET][i] = gload16(src + (size_t)poff[i] * Csrc + cc);
      inside_set[SET] = c < Csrc ? inside_lt : 0u;
    };
    // P1 of chunk period qn: prologue in place, block maximum -> slot[qn & 1]
    auto stage_max = [&](auto SETC, int qn) __attribute__((always_inline)) {
      constexpr int SET = decltype(SETC)::value;
      const int ch = qn % nch;
      const bool pro = (p.in_coef != nullptr) && ch < p.nch0;
      const unsigned msk = inside_set[SET];
      unsigned mx = 0u;
      // straight-line code (selects, one uniform branch outside the loop): hipcc then keeps counted vmcnt waits for the
      // set being consumed and leaves the other set's loads in flight
      pin(ca[SET]);
      pin(cb[SET]);
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) pin(v[SET][i]);
      if (pro) {
        const float4v a4 = ca[SET], b4 = cb[SET];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
          const bool ok = (msk >> i) & 1u;
          float4v x = v[SET][i];
          x.x = silu_fast(fmaf(a4.x, x.x, b4.x));
          x.y = silu_fast(fmaf(a4.y, x.y, b4.y));
          x.z = silu_fast(fmaf(a4.z, x.z, b4.z));
          x.w = silu_fast(fmaf(a4.w, x.w, b4.w));
          x.x = ok ? x.x : 0.f;                      // padding is exactly zero: it pads the ACTIVATED tensor
          x.y = ok ? x.y : 0.f;
          x.z = ok ? x.z : 0.f;
          x.w = ok ? x.w : 0.f;
          v[SET][i] = x;
          mx = max(max(mx, absbits(x.x)), max(max(absbits(x.y), absbits(x.z)), absbits(x.w)));
        }
      } else {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
          const bool ok = (msk >> i) & 1u;
          float4v x = v[SET][i];
          x.x = ok ? x.x : 0.f;
          x.y = ok ? x.y : 0.f;
          x.z = ok ? x.z : 0.f;
          x.w = ok ? x.w : 0.f;
          v[SET][i] = x;
          mx = max(max(mx, absbits(x.x)), max(max(absbits(x.y), absbits(x.z)), absbits(x.w)));
        }
      }
#pragma unroll
      for (int off = 32; off; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
      if (lane == 0) atomicMax(&slot[qn & 1], mx);
    };
    // P2 of chunk period qn: block scale, split into the two fp16 planes -> tile buffer qn & 1
    auto stage_write = [&](auto SETC, int qn) __attribute__((always_inline)) {
      constexpr int SET = decltype(SETC)::value;
      const int ch = qn % nch;
      const unsigned bmx = slot[qn & 1];
      if (ch == 0) e_run_p = 16;
      const int e_ch = min(max((int)(__builtin_amdgcn_readfirstlane(bmx) >> 23), 16), 254);
      e_run_p = max(e_run_p, e_ch);
      const float sc = __uint_as_float((unsigned)(268 - e_run_p) << 23);
      unsigned char* in_tile = smem + (qn & 1) * Cfg::IN_BYTES;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        if ((i + 1) * 256 <= IN_PIX * 8 || ((ptid + i * 256) >> 3) < IN_PIX) {
          const float4v xs = v[SET][i] * sc;
          const half4 h1 = __builtin_convertvector(xs, half4);
          const float4v rs = (xs - __builtin_convertvector(h1, float4v)) * 2048.f;
          const half4 h2 = __builtin_convertvector(rs, half4);
          unsigned char* dst = in_tile + wroff[i];
          *reinterpret_cast<half4*>(dst) = h1;
          *reinterpret_cast<half4*>(dst + 64) = h2;
        }
        if (i % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // (bounds the live temporaries: no spill at 256 VGPRs)
      }
    };

    // ---- producer side of the epilogue
    int ep_ct = 0, ep_b = 0, ep_ty = 0, ep_tx = 0, ep_half1 = 0;
    bool ep_pending = false, st_pending = false;
    int st_ct = 0, st_b = 0, st_ty = 0, st_tx = 0;
    float4v ep_bias = float4v{0.f, 0.f, 0.f, 0.f}, ep_osc = float4v{1.f, 1.f, 1.f, 1.f};
    float4v ep_ra = ep_bias, ep_rb = ep_bias;           // raw loads (gload16); selected by ep_use_* when they are read
    bool ep_cok = false;
    float4 ep_s1 = make_float4(0.f, 0.f, 0.f, 0.f), ep_s2 = ep_s1;
    const int pw = wave & 3;                          // the consumer whose slab this producer drains
    const int pwm = pw / WN, pwn = pw % WN;
    // constants of the tile that ends with this period: issued BEFORE the next prefetch, so that waiting for them later
    // only waits for loads that are due by then anyway (vector-memory operations retire in order)
    auto ep_load_consts = [&](int j) __attribute__((always_inline)) {
      tile_coords(j, ep_ct, ep_b, ep_ty, ep_tx);
      const int chn = (ep_ct * WN + pwn) * 64 + (lane & 15) * 4;
      const bool cok = chn < p.Cout;
      // unconditional loads from clamped addresses (gload16), selected when they are read
      const int chc = cok ? chn : 0;
      ep_cok = cok;
      ep_osc = gload16((p.oscale ? p.oscale : p.wpack) + chc);
      ep_bias = gload16((p.bias ? p.bias : p.wpack) + chc);
      ep_ra = gload16(p.res_coef ? p.res_coef + (size_t)(ep_b * 2 + 0) * p.Cout + chc : p.wpack);
      ep_rb = gload16(p.res_coef ? p.res_coef + (size_t)(ep_b * 2 + 1) * p.Cout + chc : p.wpack);
    };
    // behind the period's vm_wait: the constants become usable values
    auto ep_fix_consts = [&]() __attribute__((always_inline)) {
      pin(ep_osc);
      pin(ep_bias);
      pin(ep_ra);
      pin(ep_rb);
      const float4v zero4 = float4v{0.f, 0.f, 0.f, 0.f}, one4 = float4v{1.f, 1.f, 1.f, 1.f};
      ep_osc = (ep_cok && p.oscale) ? ep_osc : one4;
      ep_bias = (ep_cok && p.bias) ? ep_bias : zero4;
      ep_ra = (ep_cok && p.res_coef) ? ep_ra : zero4;
      ep_rb = (ep_cok && p.res_coef) ? ep_rb : zero4;
    };
    // 32 slab rows (half hb of consumer pw) -> NHWC rows
    auto ep_rows = [&](int hb, const unsigned char* half_base) __attribute__((always_inline)) {
      const float* wl = reinterpret_cast<const float*>(half_base) + pw * (32 * EP);
      const int c4e = lane & 15, rsub = lane >> 4;
      const int chn = (ep_ct * WN + pwn) * 64 + c4e * 4;
      const bool cok = chn < p.Cout;
      const int oy0 = ep_ty * TH, ox0 = ep_tx * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rrow = i * 4 + rsub;
        const int row = pwm * 64 + hb * 32 + rrow;
        const int oy = oy0 + row / 16, ox = ox0 + row % 16;
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
      }
    };
    // this wave's GroupNorm partials -> LDS (combined across waves after the next barrier, in a fixed order)
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
          float* q = red + (pw * 64 + (lane & 15) * 4) * 2;
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
    // stat tiles are 8 rows x 16 columns = the 128 pixels of two consumers along M (conv_args.h write_stats_grid)
    auto ep_stats_combine = [&]() __attribute__((always_inline)) {
      st_pending = false;
      if (!p.stats || ptid >= 128) return;
      const float* red = reinterpret_cast<const float*>(smem + Cfg::RED);
      const int q = ptid >> 6, c = ptid & 63;
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

    // ---- period -1: chunk 0 of the first tile
    issue_loads(I0{}, 0);
    issue_loads(I1{}, 1);
    vm_wait<NLOAD + 2>();
    stage_max(I0{}, 0);
    PS_BAR();
    stage_write(I0{}, 0);
    issue_loads(I0{}, 2);
    PS_BAR();

    // one chunk period q; SETN = the register set that holds chunk q + 1 (= (q + 1) & 1)
    auto period = [&](auto SETN, int q) __attribute__((always_inline)) {
      const int j = q / nch, ch = q - j * nch;
      const bool last_ch = ch == nch - 1;
      // ---- P1.  In flight, oldest first: [chunk q + 1] [constants of the tile that has just ended] [chunk q + 2]:
      // leave the NLOAD + 2 loads of chunk q + 2 in flight.  vmcnt counts STORES too, in order: every store of a period is
      // therefore issued right here, behind the wait, and is a whole period old when the next wait comes.
      vm_wait<NLOAD + 2>();
      if (st_pending) ep_stats_combine();
      if (ep_pending) {
        // the aliased slab half must be out before this period's refill of its tile buffer (P2)
        ep_fix_consts();
#ifdef DMH_STAMPS
        if (!(p.ablate & 4))
#endif
        {
          ep_rows(1, smem + ep_half1);
          ep_rows(0, smem + Cfg::SLAB0);
        }
        ep_stats_partial();                           // (combined behind the next period's wait, two barriers later)
        ep_pending = false;
      }
      PS_STAMP(0)  // wait for the halo loads; slab -> rows
#ifdef DMH_STAMPS
      if (!(p.ablate & 2))
#endif
      if (q + 1 < Q) stage_max(SETN, q + 1);
      PS_STAMP(1)  // prologue + maximum
      PS_BAR();  // M: block maximum complete
      PS_STAMP(2)  // wait at M
#ifdef DMH_STAMPS
      if (!(p.ablate & 2))
#endif
      if (q + 1 < Q) stage_write(SETN, q + 1);
      if (ptid == 0) slot[q & 1] = 0u;               // (the consumers read it before M)
      if (last_ch) ep_load_consts(j);
#ifdef DMH_STAMPS
      if (!(p.ablate & 8))
#endif
      issue_loads(SETN, q + 3);
      PS_STAMP(3)  // split + LDS write, first slab half -> rows, load issue
      PS_BAR();  // E: tile buffer (q + 1) & 1 is staged; the consumers are done with buffer q & 1
      PS_STAMP(4)  // wait at E
      if (last_ch) {
        PS_BAR();  // S: the consumers have written the second slab half
        PS_STAMP(5)  // wait at S
        ep_half1 = Cfg::ALIAS ? (q & 1) * Cfg::IN_BYTES : Cfg::SLAB1;
        ep_pending = true;
      }
    };
    for (int q = 0; q < Q; q += 2) {
      period(I1{}, q);
      if (q + 1 < Q) period(I0{}, q + 1);
    }
    // ---- tail: the last tile's rows
    if (ep_pending) {
      vm_wait<0>();
      if (st_pending) ep_stats_combine();
      ep_fix_consts();
      ep_rows(1, smem + ep_half1);
      ep_rows(0, smem + Cfg::SLAB0);
      ep_stats_partial();
    }
    PS_BAR();
    if (st_pending) ep_stats_combine();
  } else if constexpr (MF == 32) {
    // ===================================================================================================================
    //                                      CONSUMERS, v_mfma_f32_32x32x16_f16
    // ===================================================================================================================
    // A wave's 64 pixels are two 32-row blocks (= two tile rows each), its 64 output channels two 32-column blocks; a chunk's
    // K = 32 channels is two K steps of 16.  The 32x32 shape holds the SIMD's vector issue for 8 of its 32 cycles (the
    // 16x16x32 one for 8 of its 16), which is what leaves issue slots to the producer wave on the same SIMD.
    // Fragment maps: lane l, r = l & 31, h = l >> 5: A[row r][k = 8h + j], B[k = 8h + j][col r]; C: col = l & 31,
    // row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).  The packed weight keeps its 16x16x32 fragment order
    // [cout half][16-col block][plane][k group * 16 + col][8]: a 32x32x16 B fragment is read from it with per-lane addresses.
    const int r31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    int arow[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) arow[mb] = (wm * 4 + mb * 2 + (r31 >> 4)) * ROWP + (r31 & 15) * PITCH + h * 16;
    int boff[2];                                      // uint4 offset of this lane's B slice for K step ks (plane g1; g2: + 64)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) boff[ks] = (((r31 >> 4) & 1) * 2) * 64 + (2 * ks + h) * 16 + (r31 & 15);
    floatx16 acc[2][2];
    uint4 bq[NB][4];                                  // [K step][plane]
    half8 a[2][2][2][2];                              // [buffer][32-row block][K step][plane]
    int e_run = 16;
    const int nsteps = nch * NSTEP;                   // weight steps per tile
    auto wbase_of = [&](int j) __attribute__((always_inline)) {
      int ct, b, ty, tx;
      tile_coords(j < T ? j : T - 1, ct, b, ty, tx);
      return reinterpret_cast<const uint4*>(p.wpack) + (size_t)(ct * WN + wn) * nsteps * STEP_U4;
    };
    const uint4* wcur = wbase_of(0);
    const uint4* wnxt = wbase_of(1);
    auto load_b = [&](int buf, int step) __attribute__((always_inline)) {
      const uint4* src = step < nsteps ? wcur + (size_t)step * STEP_U4 : wnxt + (size_t)(step - nsteps) * STEP_U4;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) bq[buf][ks * 2 + pl] = src[boff[ks] + pl * 64];
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
    auto mma_taps = [&](auto T0C, auto T1C, const unsigned char* in_tile) __attribute__((always_inline)) {
      constexpr int T0 = decltype(T0C)::value, T1 = decltype(T1C)::value;
#pragma unroll
      for (int st = T0 * 2; st < T1 * 2; ++st) {    // st = 2 * tap + (32-column block)
        load_b((st + NB - 1) % NB, step + NB - 1);
        __builtin_amdgcn_sched_barrier(0);
        const int tap = st >> 1, cb = st & 1;
        if (cb == 0 && tap + 1 < NTAPS) read_a((tap + 1) & 1, in_tile, tap + 1);
        half8 g1s[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) g1s[ks] = __builtin_bit_cast(half8, bq[st % NB][ks * 2]) * (_Float16)(1.0f / 2048.0f);
#define DMH_TERM(pl, bexpr)                                                                       \
  _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
      acc[mb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap & 1][mb][ks][pl], bexpr, acc[mb][cb], 0, 0, 0);
        DMH_TERM(1, g1s[ks])                                                 // h2 * g1s   (smallest terms first)
        DMH_TERM(0, __builtin_bit_cast(half8, bq[st % NB][ks * 2 + 1]))      // h1 * g2
        DMH_TERM(0, __builtin_bit_cast(half8, bq[st % NB][ks * 2]))          // h1 * g1
#undef DMH_TERM
        ++step;
      }
    };
    auto slab_write = [&](int hb, unsigned char* half_base, float inv_s) __attribute__((always_inline)) {
      float* wl = reinterpret_cast<float*>(half_base) + wave * (32 * EP);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          wl[((r & 3) + 8 * (r >> 2) + 4 * h) * EP + cb * 32 + r31] = acc[hb][cb][r] * inv_s;
    };

    // ---- period -1: start the weight stream
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) load_b(i, i);
    PS_BAR();
    PS_BAR();

    for (int q = 0; q < Q; ++q) {
      const int j = q / nch, ch = q - j * nch;
      const bool last_ch = ch == nch - 1;
      unsigned char* buf = smem + (q & 1) * Cfg::IN_BYTES;
      if (ch == 0) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][cb][r] = 0.f;
        e_run = 16;
      }
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
          for (int cb = 0; cb < 2; ++cb) acc[mb][cb] *= f;
      }
#ifdef DMH_STAMPS
      if (!(p.ablate & 1))
#endif
      {
        read_a(0, buf, 0);
        mma_taps(I0{}, std::integral_constant<int, Cfg::TAPS_A>{}, buf);
      }
      PS_STAMP(0)
      PS_BAR();  // M
      PS_STAMP(2)
#ifdef DMH_STAMPS
      if (!(p.ablate & 1))
#endif
      mma_taps(std::integral_constant<int, Cfg::TAPS_A>{}, std::integral_constant<int, NTAPS>{}, buf);
      PS_STAMP(1)
      const float inv_s = __uint_as_float((unsigned)(e_run - 14) << 23);
      if (last_ch) slab_write(0, smem + Cfg::SLAB0, inv_s);
      PS_STAMP(3)
      PS_BAR();  // E
      PS_STAMP(4)
      if (last_ch) {
        slab_write(1, Cfg::ALIAS ? buf : smem + Cfg::SLAB1, inv_s);
        PS_BAR();  // S
        PS_STAMP(5)
        step = 0;
        wcur = wnxt;
        wnxt = wbase_of(j + 2);
      }
    }
    PS_BAR();  // tail: the producers' last rows
  } else {
    // ===================================================================================================================
    //                                                     CONSUMERS
    // ===================================================================================================================
    const int kg = lane >> 4, l15 = lane & 15;
    const int wm = wave / WN, wn = wave % WN;
    int arow[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) arow[mb] = (wm * 4 + mb) * ROWP + l15 * PITCH + kg * 16;
    float4v acc[4][4];
    uint4 bq[NB][4];
    half8 a[2][4][2];
    int e_run = 16;
    const int nsteps = nch * NSTEP;                   // weight steps per tile
    auto wbase_of = [&](int j) __attribute__((always_inline)) {
      int ct, b, ty, tx;
      tile_coords(j < T ? j : T - 1, ct, b, ty, tx);
      return reinterpret_cast<const uint4*>(p.wpack) + (size_t)(ct * WN + wn) * nsteps * STEP_U4 + lane;
    };
    const uint4* wcur = wbase_of(0);
    const uint4* wnxt = wbase_of(1);
    auto load_b = [&](int buf, int step) __attribute__((always_inline)) {            // step counts from the start of the current tile
      const uint4* src = step < nsteps ? wcur + (size_t)step * STEP_U4 : wnxt + (size_t)(step - nsteps) * STEP_U4;
#pragma unroll
      for (int i = 0; i < 4; ++i) bq[buf][i] = src[i * 64];
    };
    auto read_a = [&](int ab, const unsigned char* in_tile, int tap) __attribute__((always_inline)) {
      const unsigned char* at = in_tile + (tap / KW) * ROWP + (tap % KW) * PITCH;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a[ab][mb][pl] = *reinterpret_cast<const half8*>(at + arow[mb] + pl * 64);
    };
    int step = 0;                                     // weight step inside the current tile
    // taps [T0, T1) of the chunk staged in in_tile
    auto mma_taps = [&](auto T0C, auto T1C, const unsigned char* in_tile) __attribute__((always_inline)) {
      constexpr int T0 = decltype(T0C)::value, T1 = decltype(T1C)::value;
#pragma unroll
      for (int st = T0 * 2; st < T1 * 2; ++st) {
        load_b((st + NB - 1) % NB, step + NB - 1);  // weights NB - 1 steps ahead
        __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks the loads next to their use)
        const int tap = st >> 1;
        if ((st & 1) == 0 && tap + 1 < NTAPS) read_a((tap + 1) & 1, in_tile, tap + 1);   // A fragments one tap ahead
        half8 g1s[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) g1s[nb] = __builtin_bit_cast(half8, bq[st % NB][nb * 2]) * (_Float16)(1.0f / 2048.0f);
#define DMH_TERM(pl, bexpr)                                                                       \
  _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
      acc[mb][(st & 1) * 2 + nb] =                                                                \
          __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tap & 1][mb][pl], bexpr, acc[mb][(st & 1) * 2 + nb], 0, 0, 0);
        DMH_TERM(1, g1s[nb])                                                 // h2 * g1s   (smallest terms first)
        DMH_TERM(0, __builtin_bit_cast(half8, bq[st % NB][nb * 2 + 1]))      // h1 * g2
        DMH_TERM(0, __builtin_bit_cast(half8, bq[st % NB][nb * 2]))          // h1 * g1
#undef DMH_TERM
        ++step;
      }
    };
    auto slab_write = [&](int hb, unsigned char* half_base, float inv_s) __attribute__((always_inline)) {
      float* wl = reinterpret_cast<float*>(half_base) + wave * (32 * EP);
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) wl[(m2 * 16 + kg * 4 + r) * EP + nb * 16 + l15] = acc[hb * 2 + m2][nb][r] * inv_s;
    };

    // ---- period -1: start the weight stream
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) load_b(i, i);
    PS_BAR();
    PS_BAR();

    for (int q = 0; q < Q; ++q) {
      const int j = q / nch, ch = q - j * nch;
      const bool last_ch = ch == nch - 1;
      unsigned char* buf = smem + (q & 1) * Cfg::IN_BYTES;
      if (ch == 0) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = float4v{0.f, 0.f, 0.f, 0.f};
        e_run = 16;
      }
      // ---- block scale of this chunk: running maximum over the tile's chunks, so the scale only ever shrinks
      const unsigned bmx = slot[q & 1];
      const int e_old = e_run;
      const int e_ch = min(max((int)(__builtin_amdgcn_readfirstlane(bmx) >> 23), 16), 254);
      e_run = max(e_run, e_ch);
      if (e_run != e_old && ch > 0) {
        const int fe = 127 + e_old - e_run;
        const float f = fe > 0 ? __uint_as_float((unsigned)fe << 23) : 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[mb][nb] *= f;
      }
      read_a(0, buf, 0);
      mma_taps(I0{}, std::integral_constant<int, Cfg::TAPS_A>{}, buf);
      PS_STAMP(0)  // matrix work, first taps
      PS_BAR();  // M
      PS_STAMP(2)  // wait at M
      mma_taps(std::integral_constant<int, Cfg::TAPS_A>{}, std::integral_constant<int, NTAPS>{}, buf);
      PS_STAMP(1)  // matrix work, remaining taps
      const float inv_s = __uint_as_float((unsigned)(e_run - 14) << 23);  // 1 / block scale
      if (last_ch) slab_write(0, smem + Cfg::SLAB0, inv_s);
      PS_STAMP(3)  // first slab half
      PS_BAR();  // E: every consumer is done reading this tile buffer
      PS_STAMP(4)  // wait at E
      if (last_ch) {
        slab_write(1, Cfg::ALIAS ? buf : smem + Cfg::SLAB1, inv_s);
        PS_BAR();  // S
        PS_STAMP(5)  // second slab half + wait at S
        step = 0;
        wcur = wnxt;
        wnxt = wbase_of(j + 2);
      }
    }
    PS_BAR();  // tail: the producers' last rows
    }
#ifdef DMH_STAMPS
  if (dbg && lane == 0) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory");
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 6; ++i) d[i] = tk[i];
    d[6] = t_now - t_begin;
    d[7] = __builtin_amdgcn_s_memrealtime() - rt_begin;
  }
#endif
}

// ------------------------------------------------------------------------------ host side
static int ps_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int KH, int KW, int TH, int WM, int WN, int MF>
static int launch_ps(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  using Cfg = PsCfg<KH, KW, TH, WM, WN, MF>;
  ConvArgs a = fill_conv_args(d, Hout, Wout, KC, TH, 16);
  a.oscale = d->wpack + dmh_f16x3_pack_floats(d->Cout, a.C0, a.C1, KH, KW) - (int64_t)cdiv(d->Cout, 64) * 64;
  auto kern = conv_f16x3_ps_kernel<KH, KW, TH, WM, WN, MF>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Cfg::LDS_BYTES);
    DMH_REQUIRE(e == hipSuccess, "dmh_conv2d: cannot raise the LDS limit: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int ntiles = a.tilesX * a.tilesY * a.B * cdiv(a.Cout, 64 * WN);
#ifdef DMH_STAMPS
  if (const char* e = getenv("DMH_WINO_ABLATE")) a.ablate = atoi(e);
#endif
  int grid = ps_num_cus() & ~7;                      // one persistent workgroup per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
#ifdef DMH_STAMPS
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, st, a, ntiles, d->upsample2 == 1 ? 1 : 0, g_ps_dbg);
#else
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, st, a, ntiles, d->upsample2 == 1 ? 1 : 0);
#endif
  DMH_CHECK_LAUNCH("dmh_conv2d(f16x3 producer/consumer)");
  return DMH_OK;
}

// Which launches take the specialised kernel (DMH_CONV_PS: 0 = never, 1 = the measured default, 2 = wherever it can run)
bool dmh_f16x3_ps_wanted(const DmhConv* d, int Hout, int Wout) {
  static const int mode = [] {
    const char* e = getenv("DMH_CONV_PS");
    return e ? atoi(e) : 1;
  }();
  if (mode == 0) return false;
  const int C1 = d->src1 ? d->C1 : 0;
  const int nch = cdiv(d->C0, KC) + cdiv(C1, KC);
  if (!(d->stride == 1 && (d->KH == 3 || d->KH == 1) && d->KH == d->KW && d->upsample2 == 0 && nch >= 2)) return false;
  if (mode >= 2) return true;
  return d->KH == 3;
}

int dmh_f16x3_ps_launch(const DmhConv* d, int Hout, int Wout, hipStream_t st) {
  const bool wide = d->Cout % 128 == 0;
  static const int mf = [] {                           // development knob: 16 = v_mfma_f32_16x16x32_f16 consumers
    const char* e = getenv("DMH_PS_MFMA");
    return e ? atoi(e) : 32;
  }();
  if (mf == 16) {
    if (d->KH == 1)
      return wide ? launch_ps<1, 1, 8, 2, 2, 16>(d, Hout, Wout, st) : launch_ps<1, 1, 16, 4, 1, 16>(d, Hout, Wout, st);
    return wide ? launch_ps<3, 3, 8, 2, 2, 16>(d, Hout, Wout, st) : launch_ps<3, 3, 16, 4, 1, 16>(d, Hout, Wout, st);
  }
  if (d->KH == 1)
    return wide ? launch_ps<1, 1, 8, 2, 2, 32>(d, Hout, Wout, st) : launch_ps<1, 1, 16, 4, 1, 32>(d, Hout, Wout, st);
  return wide ? launch_ps<3, 3, 8, 2, 2, 32>(d, Hout, Wout, st) : launch_ps<3, 3, 16, 4, 1, 32>(d, Hout, Wout, st);
}
